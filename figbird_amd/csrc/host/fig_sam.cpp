// fig_sam.cpp -- see fig_sam.h.  Restates Preprocess.cpp's decisions (citations are /root/reference/Preprocess.cpp lines).
// Where the reference prints heap garbage -- `md` of records without an MD tag (unmapped mates, :1531-1545) and `ih` of
// improperly paired records (never set before writeSam, :404-410) -- this file prints "" and 1; Figbird.cpp reads neither
// (parseUnmapped takes columns 2, 4 of the anchor line and column 7 of the mate line, Figbird.cpp:5700-5736).
#include "fig_sam.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace figsam {
namespace {

const int kRec = 1024;                  // MAX_REC_LEN
const int kReadCap = 3000;
const long kMaxFragment = 5000;         // MAX_FRAGMENT_SIZE

struct Sam {
    std::string qname; int flag = 0; std::string rname; int pos = 0, mapq = 0; std::string cigar, rnext; int pnext = 0, tlen = 0;
    std::string seq, qual, md; long ih = 1; int nm = -1, as = 0; long contigNo = -1;
};

struct State {
    Args a;
    std::vector<std::string> contigs, contigNames;
    std::unordered_map<std::string, long> nameIndex;       // getContigNo (:327-338): first contig of that name wins
    std::vector<GapRec> gaps;
    // per-gap bookkeeping
    std::vector<int> read_count, partial_read_count;
    std::vector<std::vector<std::string>> jump_reads;       // unmapped_jump_reads
    std::vector<std::unordered_set<std::string>> jump_set, partial_set;
    std::vector<std::string> gap_text, partial_text;
    std::vector<int> perfect_gap, perfect_len;
    // interval index: per contig the gap ids in file order, plus whether starts/ends ascend (then binary search applies)
    std::vector<std::vector<int>> by_contig;
    std::vector<char> sorted_contig;
    long totalCount = 0, unCount = 0; unsigned long maxReadLength = 0;
    int read_mean = 0;
    int cigar_val[3] = {0, 0, 0};
    std::string myout;                  // text of myout.sam not yet written (streamed out in ~1 MB pieces)
    FILE *myout_file = nullptr; bool myout_failed = false;
    bool account = false;               // first pass of a far jump library: every line written also feeds the insert-size histogram
    std::vector<long> insertCounts; long discarded = 0;
    FILE *out1 = nullptr, *out2 = nullptr; bool writeflag = false;
};

// processMapping over a line of myout.sam as the reference re-reads it (:1768-1830): strtok on tabs (empty fields vanish),
// insert size = 6th token, MD / IH from the tokens behind the 7th
void account_myout_line(State &S, const char *p, size_t n) {
    const char *f[16]; size_t fl[16]; int nf = 0;
    size_t q = 0;
    while (q <= n && nf < 16) { size_t t = q; while (t < n && p[t] != '\t') t++; if (t > q) { f[nf] = p + q; fl[nf] = t - q; nf++; } q = t + 1; }
    if (nf < 7) return;
    const int insertSize = atoi(std::string(f[5], fl[5]).c_str());
    const char *md = nullptr; size_t mdl = 0; int nh = 0;
    for (int k = 7; k < nf; k++) {
        if (fl[k] >= 2 && f[k][0] == 'M' && f[k][1] == 'D') { md = f[k]; mdl = fl[k]; }
        else if (fl[k] >= 2 && f[k][0] == 'I' && f[k][1] == 'H') nh = fl[k] > 5 ? atoi(std::string(f[k] + 5, fl[k] - 5).c_str()) : 0;
    }
    if (nh == 1 && !(mdl > 5 && md[5] == '^')) {
        const long c = atol(std::string(f[2], fl[2]).c_str());
        if (c >= 0 && c < (long)S.contigs.size() && !S.contigs[(size_t)c].empty()) {
            if (insertSize > 0) { if (insertSize < kMaxFragment) S.insertCounts[(size_t)insertSize]++; else if (insertSize > kMaxFragment) S.discarded++; }
        }
    }
}

// a record for myout.sam: appended to the pending text, accounted when asked, and the text handed to the file in pieces
void myout_line(State &S, const Sam &r);

std::string revcomp(const std::string &s) {     // reverse(), :145-166
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) {
        char ch = s[i], o = 'N';
        if (ch == 'A') o = 'T'; else if (ch == 'C') o = 'G'; else if (ch == 'G') o = 'C'; else if (ch == 'T') o = 'A';
        r[s.size() - 1 - i] = o;
    }
    return r;
}

bool get_sam(State &S, char *line, Sam &sam) {     // getSAM, :1491-1551
    sam = Sam();
    char *sv = nullptr;
    const char *d = "\t\n ";
    char *t = strtok_r(line, d, &sv); if (!t) return false; sam.qname = t;
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.flag = atoi(t);
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.rname = t;
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.pos = atoi(t);
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.mapq = atoi(t);
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.cigar = t;
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.rnext = t;
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.pnext = atoi(t);
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.tlen = atoi(t);
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.seq = t;
    t = strtok_r(nullptr, d, &sv); if (!t) return false; sam.qual = t;
    while ((t = strtok_r(nullptr, d, &sv)) != nullptr) {
        if (t[0] == 'M' && t[1] == 'D') sam.md = t;
        if (t[0] == 'N' && t[1] == 'M') sam.nm = strlen(t) > 5 ? atoi(t + 5) : 0;
        if (t[0] == 'A' && t[1] == 'S') sam.as = strlen(t) > 5 ? atoi(t + 5) : 0;
    }
    auto it = S.nameIndex.find(sam.rname);
    sam.contigNo = it == S.nameIndex.end() ? -1 : it->second;
    return true;
}

void sam_line(std::string &dst, const Sam &r) {     // writeSam / writeSam2, :404-417
    char buf[64];
    dst += r.qname; dst += '\t';
    snprintf(buf, sizeof buf, "%d\t%ld\t%d\t", r.flag, r.contigNo, r.pos); dst += buf;
    dst += r.cigar; dst += '\t';
    snprintf(buf, sizeof buf, "%d\t", r.tlen); dst += buf;
    dst += r.seq; dst += '\t'; dst += r.qual; dst += '\t'; dst += r.md;
    snprintf(buf, sizeof buf, "\tIH:i:%ld\n", r.ih); dst += buf;
}

void myout_line(State &S, const Sam &r) {
    const size_t at = S.myout.size();
    sam_line(S.myout, r);
    if (S.account) account_myout_line(S, S.myout.data() + at, S.myout.size() - at - 1);
    if (S.myout.size() >= (1u << 20)) {
        if (S.myout_file && fwrite(S.myout.data(), 1, S.myout.size(), S.myout_file) != S.myout.size()) S.myout_failed = true;
        S.myout.clear();
    }
}

int parse_del(const std::string &cigar) {           // parseDel, :168-200: leading soft clip length (S before the first M)
    size_t p1 = cigar.find('S'), p2 = cigar.find('M');
    if (p1 != std::string::npos && p2 != std::string::npos && p1 < p2) return atoi(cigar.substr(0, p1).c_str());
    return 0;
}

void parse_cigar(State &S, const std::string &cigar, int readlen) {      // parse_Cigar, :202-290 (additive: callers reset cigar_val)
    size_t p1 = cigar.find('S'), p2 = cigar.find('M');
    if (p1 == std::string::npos || p2 == std::string::npos || !(p1 < p2)) return;
    S.cigar_val[0] = atoi(cigar.substr(0, p1).c_str());
    S.cigar_val[1] = atoi(cigar.substr(p1 + 1, p2 - p1 - 1).c_str());
    if (S.cigar_val[0] + S.cigar_val[1] == readlen) return;
    std::string s = cigar.substr(p2 + 1);
    std::vector<int> s_index;
    for (size_t i = 0; i < s.size(); i++) if (s[i] == 'S') s_index.push_back((int)i);
    if (s_index.empty()) return;
    auto has_op = [](const std::string &x) { return x.find_first_of("DIMX=") != std::string::npos; };
    if (has_op(s)) {
        const int last = s_index.back();
        std::string t;
        for (int k = last - 2; k < last; k++) t.push_back(k >= 0 && k < (int)s.size() ? s[k] : '\0');   // (k < 0 reads before the buffer in the reference)
        { size_t z = t.find('\0'); if (z != std::string::npos) t.resize(z); }
        if (has_op(t)) { t.clear(); if (last - 1 >= 0) t.push_back(s[last - 1]); }
        S.cigar_val[2] = atoi(t.c_str());
    } else S.cigar_val[2] = readlen - S.cigar_val[0] - S.cigar_val[1];
}

bool check_char(const std::string &r) {             // checkChar, :868-883: true = has a character outside ACGTNacgtn
    for (char c : r) if (!strchr("ACGTNacgtn", c)) return true;
    return false;
}
bool ncount_ok(const std::string &r) { int c = 0; for (char ch : r) if (ch == 'N') c++; return c <= 3; }      // check_Ncount_partial, :857-866

bool dup_jump(State &S, const std::string &read, int g) {      // check_duplicate, samflag 2 (:369-387)
    if (S.jump_set[g].count(read)) return true;
    if (S.read_count[g] == 0) return false;
    const std::string core = read.size() >= 4 ? read.substr(2, read.size() - 4) : std::string();       // clip 2 from either end
    // (memmem: glibc's vectorised two-way search; this scan over every read already kept for the gap is the reference's own
    //  quadratic duplicate test and dominates the ingest at thousands of reads per gap)
    if (core.empty()) return !S.jump_reads[g].empty();
    for (const std::string &s1 : S.jump_reads[g]) if (s1.size() >= core.size() && memmem(s1.data(), s1.size(), core.data(), core.size())) return true;
    return false;
}

void check_mim(State &S, const std::string &cigar, int g) {    // checkMIM, :885-925
    int index1 = 0, index2 = 0, index3 = 0, m_count = 0, i_count = 0;
    for (int i = 0; i < (int)cigar.size(); i++) {
        const char c = cigar[i];
        if (c == 'S' || c == 'D' || c == '=' || c == 'X') return;
        if (c == 'M') { if (m_count == 0) index1 = i; else if (m_count == 1) index3 = i; else return; m_count++; }
        else if (c == 'I') { if (i_count == 1) return; index2 = i; i_count++; }
    }
    if (index1 && index2 && index3 && index1 < index2 && index2 < index3) {
        S.perfect_gap[g] = 1;
        S.perfect_len[g] = atoi(cigar.substr(index1 + 1, index2 - index1 - 1).c_str()) + 1;
    }
}

// ---- gap lookup through the per-contig index ---------------------------------------------------------------------------
int check_range(int a, int b, int mean) {            // checkRange, :526-534
    int lo = mean - 1000, hi = mean + 1000;
    if (a > lo && a < hi) return 1;
    if (b > lo && b < hi) return 1;
    if ((a < lo && b > hi) || (b < lo && a > hi)) return 1;
    return 0;
}
int check_insert(int a, int b, double mean) { return std::fabs(mean - a) < std::fabs(mean - b) ? a : b; }     // checkInsert, :516-524

template <class F> void for_gaps_of(const State &S, long contigNo, F f) {       // candidates in the reference's scan order
    if (contigNo < 0 || contigNo >= (long)S.by_contig.size()) return;
    for (int g : S.by_contig[(size_t)contigNo]) if (!f(g)) return;
}

int check_pos(State &S, long contigNo, long pos, int strandNo, int readlength) {      // checkPos, :536-614
    const int maxD = S.a.maxDistance;
    if (contigNo < 0 || contigNo >= (long)S.by_contig.size()) return -1;
    const std::vector<int> &ids = S.by_contig[(size_t)contigNo];
    // gaps whose window holds pos: strand 0  gapStart in (pos, pos+maxD);  strand 1  gapEnd in (pos-maxD, pos)
    size_t lo = 0, hi = ids.size();
    if (S.sorted_contig[(size_t)contigNo]) {
        if (strandNo == 0) {
            lo = std::partition_point(ids.begin(), ids.end(), [&](int g) { return S.gaps[g].gapStart <= pos; }) - ids.begin();
            hi = std::partition_point(ids.begin(), ids.end(), [&](int g) { return S.gaps[g].gapStart < pos + maxD; }) - ids.begin();
        } else {
            lo = std::partition_point(ids.begin(), ids.end(), [&](int g) { return S.gaps[g].gapStart + S.gaps[g].gapLength <= pos - maxD; }) - ids.begin();
            hi = std::partition_point(ids.begin(), ids.end(), [&](int g) { return S.gaps[g].gapStart + S.gaps[g].gapLength < pos; }) - ids.begin();
        }
    }
    int flag = 0, gap_index = 0, min_val = 1000000, min_index = -1;
    std::vector<std::pair<int, int>> tis;              // (gap, tempinsertsize) of the matches
    for (size_t k = lo; k < hi; k++) {
        const int i = ids[k];
        const GapRec &g = S.gaps[i];
        const bool m = (strandNo == 0 && pos > g.gapStart - maxD && pos < g.gapStart) ||
                       (strandNo == 1 && pos > g.gapStart + g.gapLength && pos < g.gapStart + g.gapLength + maxD);
        if (!m) continue;
        if (maxD <= 250) return i;
        int val0, val1, t = 0;
        if (pos < g.gapStart) { val0 = (int)(g.gapStart + g.gapLength - pos + readlength); val1 = (int)(g.gapStart - pos + 1); }
        else { val0 = (int)(pos - g.gapStart + 2 * readlength - 1); val1 = (int)(pos - g.gapStart - g.gapLength + readlength + 1); }
        if (check_range(val0, val1, S.read_mean)) t = check_insert(val0, val1, S.read_mean);
        if (t != 0) { flag++; gap_index = i; }
        const int ab = std::abs(S.read_mean - t);
        if (ab < min_val) { min_val = ab; min_index = i; }
        tis.emplace_back(i, t);
    }
    if (maxD <= 250) return -1;
    if (flag == 0) return -1;
    const int min_thresh = (int)(S.read_mean - S.read_mean * 0.6);
    const int c_index = flag == 1 ? gap_index : min_index;
    int tc = 0;
    for (auto &p : tis) if (p.first == c_index) tc = p.second;
    return tc < min_thresh ? -1 : c_index;
}

int check_pos2(State &S, long contigNo, long pos, int readlength, int del) {          // checkPos2, :616-639 (both strands alike)
    if (contigNo < 0 || contigNo >= (long)S.by_contig.size()) return -1;
    const std::vector<int> &ids = S.by_contig[(size_t)contigNo];
    auto match = [&](int i) {
        const GapRec &g = S.gaps[i];
        const long gapEnd = g.gapStart + g.gapLength;
        return (pos > g.gapStart - readlength + 1 && pos <= g.gapStart) || (pos > gapEnd && del && (pos - del) <= gapEnd);
    };
    if (!S.sorted_contig[(size_t)contigNo]) { for (int i : ids) if (match(i)) return i; return -1; }
    int best = -1;
    size_t k = std::partition_point(ids.begin(), ids.end(), [&](int g) { return S.gaps[g].gapStart < pos; }) - ids.begin();
    if (k < ids.size() && match(ids[k])) best = ids[k];
    if (del) {
        size_t q = std::partition_point(ids.begin(), ids.end(), [&](int g) { return S.gaps[g].gapStart + S.gaps[g].gapLength < pos - del; }) - ids.begin();
        if (q < ids.size() && match(ids[q]) && (best < 0 || ids[q] < best)) best = ids[q];
    }
    return best;
}

int write_partial(State &S, const Sam &read, int gapNo, int strandNo, int del, int pos2) {   // writePartialSam, :425-502
    const int gap_s = (int)S.gaps[gapNo].gapStart, gap_e = (int)(S.gaps[gapNo].gapStart + S.gaps[gapNo].gapLength);
    const int readlength = (int)read.seq.size();
    int match = -1;
    char buf[96];
    auto emit = [&](int clipped_index) {
        std::string &dst = S.partial_text[gapNo];
        dst += read.seq;
        snprintf(buf, sizeof buf, "\t%d\t%d\t%d\t", clipped_index, match, read.pos); dst += buf;
        dst += read.cigar;
        snprintf(buf, sizeof buf, "\t%d\t", pos2); dst += buf;
        dst += read.qual; dst += '\n';
    };
    if (read.pos < gap_s) {
        match = strandNo == 0 ? 1 : 4;
        S.cigar_val[0] = S.cigar_val[1] = S.cigar_val[2] = 0;
        parse_cigar(S, read.cigar, readlength);
        if (S.cigar_val[0]) { if (S.cigar_val[2]) emit(readlength - S.cigar_val[2] - 1); }     // S..M..S kept; S..M alone dropped
        else emit(gap_s - read.pos);
    } else if (read.pos > gap_s) {
        parse_cigar(S, read.cigar, readlength);
        match = strandNo == 0 ? 2 : 3;
        emit(gap_e - 1 - read.pos + del + 2);
    }
    return match;
}

void collect_partial(State &S, const Sam &read, int pos2, int which_case) {          // collectPartialSAM :1667-1694 / case 2 of printMixedVectors :1343-1361
    (void)which_case;
    const int strandNo = (read.flag & 16) >> 4;
    const int del = parse_del(read.cigar);
    const int g = check_pos2(S, read.contigNo, read.pos, (int)read.seq.size(), del);
    if (g >= 0 && S.partial_read_count[g] <= kReadCap) {
        if (ncount_ok(read.seq) && !S.partial_set[g].count(read.seq)) {
            write_partial(S, read, g, strandNo, del, pos2);
            S.partial_set[g].insert(read.seq);
            S.partial_read_count[g]++;
            check_mim(S, read.cigar, g);
        }
    }
}

void rewrite_readset(State &S, Sam &r1, Sam &r2) {       // reWriteReadset, :1696-1732 (reverses qual IN PLACE: later writers see it)
    auto one = [](FILE *f, Sam &r) {
        if ((r.flag & 16) >> 4) {
            std::string t = revcomp(r.seq);
            std::reverse(r.qual.begin(), r.qual.end());
            fprintf(f, "@%s\n%s\n+\n%s\n", r.qname.c_str(), t.c_str(), r.qual.c_str());
        } else fprintf(f, "@%s\n%s\n+\n%s\n", r.qname.c_str(), r.seq.c_str(), r.qual.c_str());
    };
    one(S.out1, r1); one(S.out2, r2);
}

double n_frac(const std::string &s) { int c = 0; for (char ch : s) if (ch == 'N' || ch == 'n') c++; return c / (double)s.size(); }

void print_vectors(State &S, std::vector<Sam> &reads1, std::vector<Sam> &reads2) {     // printVectors, :641-855
    const unsigned long ih = reads1.size();
    for (unsigned long i = 0; i < ih; i++) {
        Sam &r1 = reads1[i], &r2 = reads2[i];
        if (r1.rname == "*" || r2.rname == "*") {
            if (r1.seq.size() > S.maxReadLength) S.maxReadLength = r1.seq.size();
            if (r2.seq.size() > S.maxReadLength) S.maxReadLength = r2.seq.size();
            if (n_frac(r1.seq) < 0.8 && n_frac(r2.seq) < 0.8) { S.unCount++; S.totalCount++; reads1.clear(); reads2.clear(); return; }
        } else if (r1.rname != r2.rname) {
        } else {
            if (r1.seq.size() > S.maxReadLength) S.maxReadLength = r1.seq.size();
            r1.ih = (long)ih;
            if (r2.seq.size() > S.maxReadLength) S.maxReadLength = r2.seq.size();
            r2.ih = (long)ih;
            myout_line(S, r1); myout_line(S, r2);
        }
    }
    S.totalCount++;
    reads1.clear(); reads2.clear();
}

void store_jump(State &S, int g, const std::string &seq) { S.jump_reads[g].push_back(seq); S.jump_set[g].insert(seq); S.read_count[g]++; }

void print_mixed(State &S, std::vector<Sam> &m1, std::vector<Sam> &m2) {              // printMixedVectors, :999-1489
    const int maxD = S.a.maxDistance, samflag = S.a.samflag;
    for (size_t oi = 0; oi < m1.size(); oi++) {
        for (size_t oj = 0; oj < m2.size(); oj++) {
            Sam &read1 = m1[oi], &read2 = m2[oj];
            if (oi == 0 && oj == 0) {
                if (read1.seq.size() > S.maxReadLength) S.maxReadLength = read1.seq.size();
                if (read2.seq.size() > S.maxReadLength) S.maxReadLength = read2.seq.size();
                if (n_frac(read1.seq) < 0.8 && n_frac(read2.seq) < 0.8) { S.unCount++; S.totalCount++; }
                else { m1.clear(); m2.clear(); return; }
            }
            if ((read1.flag & 4) != 0 && (read2.flag & 4) != 0) { m1.clear(); m2.clear(); return; }           // both unmapped
            if (((read1.flag & 4) == 0 && (read2.flag & 4) != 0) || ((read1.flag & 4) == 0 && (read2.flag & 4) == 0 && maxD > 250)) {
                const bool r2_unmapped = (read2.flag & 4) != 0;
                for (size_t i = 0; i < m1.size(); i++) {
                    const long contigNo1 = m1[i].contigNo; const long pos1 = m1[i].pos; const int strandNo1 = (m1[i].flag & 16) >> 4;
                    if (samflag == 2 && !check_char(m2[0].seq)) {
                        if (r2_unmapped) {                     // mate 2 unmapped: the pairs Figbird fills gaps with (:1202-1249)
                            const int g = check_pos(S, contigNo1, pos1, strandNo1, (int)m2[0].seq.size());
                            if (g >= 0 && S.read_count[g] <= kReadCap) {
                                const std::string temp = revcomp(m2[0].seq);
                                if ((strandNo1 == 1 && !dup_jump(S, m2[0].seq, g)) || (strandNo1 == 0 && !dup_jump(S, temp, g))) {
                                    sam_line(S.gap_text[g], m1[i]); sam_line(S.gap_text[g], m2[0]);
                                    if (strandNo1 == 0) m2[0].seq = temp;
                                    store_jump(S, g, m2[0].seq);
                                }
                            }
                        } else {                               // both mapped, improperly, jump library (:1251-1338)
                            const std::string original2 = m2[0].seq, original1 = m1[i].seq;
                            const long contigNo2 = m2[0].contigNo; const long pos2 = m2[0].pos; const int strandNo2 = (m2[0].flag & 16) >> 4;
                            int g = check_pos(S, contigNo1, pos1, strandNo1, (int)m2[0].seq.size());
                            if (g >= 0 && S.read_count[g] <= kReadCap) {
                                const std::string temp = revcomp(m2[0].seq);
                                if (strandNo2 == 1) m2[0].seq = temp;
                                const std::string rev = revcomp(m2[0].seq);
                                if ((strandNo1 == 1 && !dup_jump(S, m2[0].seq, g)) || (strandNo1 == 0 && !dup_jump(S, rev, g))) {
                                    sam_line(S.gap_text[g], m1[i]); sam_line(S.gap_text[g], m2[0]);
                                    if (strandNo1 == 0) m2[0].seq = rev;
                                    store_jump(S, g, m2[0].seq);
                                }
                            }
                            m2[0].seq = original2; m1[i].seq = original1;
                            g = check_pos(S, contigNo2, pos2, strandNo2, (int)m1[i].seq.size());
                            if (g >= 0 && S.read_count[g] <= kReadCap) {
                                const std::string temp = revcomp(m1[i].seq);
                                if (strandNo1 == 1) m1[i].seq = temp;
                                const std::string rev = revcomp(m1[i].seq);
                                if ((strandNo2 == 1 && !dup_jump(S, m1[i].seq, g)) || (strandNo2 == 0 && !dup_jump(S, rev, g))) {
                                    sam_line(S.gap_text[g], m2[0]); sam_line(S.gap_text[g], m1[i]);
                                    if (strandNo2 == 0) m1[i].seq = rev;
                                    store_jump(S, g, m1[i].seq);
                                }
                            }
                        }
                    }
                    if (samflag == 1) collect_partial(S, m1[i], -1, 1);
                }
                m1.clear(); m2.clear();
                return;
            }
            // (mate 1 unmapped / mate 2 mapped is commented out in the reference, :1381-1463: such pairs are not used)
        }
    }
    m1.clear(); m2.clear();
}

// FASTA as Preprocess.cpp reads it (:2017-2084): 1024-byte fgets pieces, last character of a short piece dropped
bool load_contigs(const std::string &path, std::vector<std::string> &contigs, std::vector<std::string> *names) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    std::vector<char> line(kRec);
    std::string cur;
    while (fgets(line.data(), kRec, f) != nullptr) {
        if (line[0] == ';') continue;
        size_t n = strlen(line.data());
        if (line[0] == '>') {
            if (names) {
                std::string nm(line.data() + 1, n >= 2 ? n - 2 : 0);
                size_t b = nm.find_first_not_of(" \t\n");
                std::string tok;
                if (b != std::string::npos) { size_t e = nm.find_first_of(" \t\n", b); tok = nm.substr(b, e == std::string::npos ? std::string::npos : e - b); }
                names->push_back(tok);
            }
            if (!cur.empty()) { contigs.push_back(cur); cur.clear(); }
        } else cur.append(line.data(), (long)n < kRec - 1 ? n - 1 : n);
    }
    contigs.push_back(cur);
    fclose(f);
    return true;
}

}  // namespace

int preprocess(const Args &a, Binned &out, std::string &err, int write_mode) {
    State S; S.a = a;
    // ---- genome reduction: gap ordinal -> contig index of the UNREDUCED genome (:1883-2007)
    std::map<int, int> contignums;
    if (a.genome_reduction == 1) {
        std::vector<std::string> full;
        if (!load_contigs(a.filledContigFile, full, nullptr)) { err = "Can't open contig file"; return 1; }
        int nStart = 0, nCount = 0, gapcount = 0;
        for (size_t i = 0; i < full.size(); i++)
            for (long j = 0; j < (long)full[i].size(); j++) {
                const bool c = full[i][j] == 'N' || full[i][j] == 'n';
                if (c) { if (nStart == 0) { nStart = 1; nCount = 1; } else nCount++; }
                if ((!c && nStart == 1) || (c && j == (long)full[i].size() - 1)) { if (nCount >= 1) { contignums.insert({gapcount, (int)i}); gapcount++; } nStart = 0; }
            }
    }
    if (!load_contigs(a.contigFile, S.contigs, &S.contigNames)) { err = "Can't open contig file"; return 1; }
    for (size_t i = 0; i < S.contigs.size() && i < S.contigNames.size(); i++) S.nameIndex.emplace(S.contigNames[i], (long)i);
    // ---- gaps (:2098-2154): a run that reaches a contig's end is recorded only when the NEXT contig's first non-N base is seen (quirk, kept)
    std::string gapInfo;
    {
        int nStart = 0, nCount = 0, gapcount = 0; long nStartPos = 0;
        for (size_t i = 0; i < S.contigs.size(); i++)
            for (long j = 0; j < (long)S.contigs[i].size(); j++) {
                const char ch = S.contigs[i][j];
                if (ch == 'N' || ch == 'n') { if (nStart == 0) { nStart = 1; nCount = 1; nStartPos = j; } else nCount++; }
                else if (nStart == 1) {
                    if (nCount >= 1) {
                        GapRec g; g.contigNo = (int)i; g.gapStart = nStartPos; g.gapLength = nCount; g.contigToWrite = (int)i;
                        if (a.genome_reduction == 1) { auto it = contignums.find(gapcount); if (it != contignums.end()) g.contigToWrite = it->second; }
                        S.gaps.push_back(g);
                        char buf[96]; snprintf(buf, sizeof buf, "%d\t%ld\t%d\n", g.contigToWrite, g.gapStart, g.gapLength); gapInfo += buf;
                        gapcount++;
                    }
                    nStart = 0;
                }
            }
    }
    const size_t ng = S.gaps.size();
    S.read_count.assign(ng, 0); S.partial_read_count.assign(ng, 0); S.jump_reads.assign(ng, {}); S.jump_set.assign(ng, {}); S.partial_set.assign(ng, {});
    S.gap_text.assign(ng, ""); S.partial_text.assign(ng, ""); S.perfect_gap.assign(ng, 0); S.perfect_len.assign(ng, 0);
    S.by_contig.assign(S.contigs.size(), {}); S.sorted_contig.assign(S.contigs.size(), 1);
    for (size_t g = 0; g < ng; g++) {
        std::vector<int> &v = S.by_contig[(size_t)S.gaps[g].contigNo];
        if (!v.empty()) { const GapRec &p = S.gaps[v.back()], &q = S.gaps[g]; if (q.gapStart < p.gapStart || q.gapStart + q.gapLength < p.gapStart + p.gapLength) S.sorted_contig[(size_t)q.contigNo] = 0; }
        v.push_back((int)g);
    }
    // ---- reduced read files (:2217-2258)
    const int def = a.default_setting;
    if (def == 1) { if (a.read_reduction == 1) S.writeflag = true; }
    else if (a.read_reduction == 1 && a.samflag == 1) S.writeflag = true;
    if (S.writeflag) {
        auto red = [](const std::string &p, const std::string &first) {
            size_t f = p.find_last_of('.'); size_t f1 = first.find_last_of('.');
            std::string base = p.substr(0, f);
            std::string stem1 = first.substr(0, f1);
            std::string ext = f1 == std::string::npos ? std::string() : first.substr(f1, stem1.size());      // s1.substr(found1, s3.size()): both files get the FIRST file's extension
            return base + "_reduced" + ext;
        };
        const std::string s3 = red(a.reads1, a.reads1), s4 = red(a.reads2, a.reads1);
        S.out1 = fopen(s3.c_str(), "w"); S.out2 = fopen(s4.c_str(), "w");
        if (!S.out1 || !S.out2) { err = "Can't create reduced read pair during preproscessing...exiting."; return 1; }
        out.stdout_lines.push_back(s3); out.stdout_lines.push_back(s4);
    }
    FILE *mapFile = fopen(a.mapFile.c_str(), "r");
    if (!mapFile) { err = "Can't open alignment file"; return 1; }
    std::vector<char> line(kRec);
    std::vector<Sam> reads1, reads2, mixed1, mixed2;
    const bool jump_far = a.samflag == 2 && a.maxDistance > 250;
    std::string preq1 = "*", preq2 = "*";
    auto next_line = [&]() { return fgets(line.data(), kRec, mapFile) != nullptr; };
    if (write_mode) {
        S.myout_file = fopen(a.outFile.c_str(), "w");
        if (!S.myout_file) { err = "Can't create myout file"; return 1; }
    }
    if (jump_far) {
        // ---- first pass (:2278-2435): properly paired records -> myout.sam, then the mean insert size read_mean
        // (the reference re-reads the file it has just written, processMapping :1768-1830; here each line is accounted as it is written)
        S.account = true; S.insertCounts.assign((size_t)kMaxFragment, 1); S.discarded = 0;
        bool end = false;
        while (next_line()) {
            if (line[0] == '@') continue;
            Sam read1; if (!get_sam(S, line.data(), read1)) continue;
            while ((read1.flag & 2) == 0) {
                for (int seg_no = 0; seg_no < 2 && !end; seg_no++) {
                    const std::string q = read1.qname; const int seg = read1.flag & 192;
                    while (q == read1.qname && (read1.flag & 192) == seg) {
                        if (!next_line() || !get_sam(S, line.data(), read1)) { end = true; break; }
                    }
                }
                if (end) break;
            }
            if (end) break;
            if (!next_line()) break;
            Sam read2; if (!get_sam(S, line.data(), read2)) break;
            if (read1.qname != preq1 || read2.qname != preq2) { preq1 = read1.qname; preq2 = read2.qname; print_vectors(S, reads1, reads2); }
            reads1.push_back(std::move(read1)); reads2.push_back(std::move(read2));
        }
        print_vectors(S, reads1, reads2);
        S.account = false;
        long insCount = S.discarded; double sum = 0;
        for (long i = 0; i < kMaxFragment; i++) { insCount += S.insertCounts[(size_t)i] - 1; sum += i * (S.insertCounts[(size_t)i] - 1); }
        S.read_mean = (int)(sum / insCount);
        fclose(mapFile);
        mapFile = fopen(a.mapFile.c_str(), "r");
        if (!mapFile) { err = "Can't open map file"; return 1; }
        preq1 = "*"; preq2 = "*";
    }
    // ---- main pass (:2437-2566)
    {
        bool end = false;
        while (next_line()) {
            if (line[0] == '@') continue;
            Sam read1; if (!get_sam(S, line.data(), read1)) continue;
            while ((read1.flag & 2) == 0) {
                {   const std::string q = read1.qname; const int seg = read1.flag & 192;
                    while (q == read1.qname && (read1.flag & 192) == seg) {
                        mixed1.push_back(read1);
                        if (!next_line() || !get_sam(S, line.data(), read1)) { end = true; break; }      // (the reference spins here forever: a SAM never ends on a lone mate)
                    } }
                if (end) { mixed1.clear(); break; }
                {   const std::string q = read1.qname; const int seg = read1.flag & 192;
                    while (q == read1.qname && (read1.flag & 192) == seg) {
                        mixed2.push_back(read1);
                        if (!next_line() || !get_sam(S, line.data(), read1)) { end = true; break; }
                    } }
                if (S.writeflag) rewrite_readset(S, mixed1[0], mixed2[0]);
                print_mixed(S, mixed1, mixed2);
                if (end) break;
            }
            if (end) break;
            if (!next_line()) break;
            Sam read2; if (!get_sam(S, line.data(), read2)) break;
            if (read1.qname != preq1 || read2.qname != preq2) {
                preq1 = read1.qname; preq2 = read2.qname;
                if (!jump_far) print_vectors(S, reads1, reads2);
                if (a.samflag == 1) {
                    const bool f1 = read1.cigar == "101M", f2 = read2.cigar == "101M";        // hard-coded full-map test (:1855-1856, :2546)
                    if (!(f1 && f2)) {
                        collect_partial(S, read1, read2.pos, 0);
                        collect_partial(S, read2, read1.pos, 0);
                        if (S.writeflag) rewrite_readset(S, read1, read2);
                    }
                }
                if (!jump_far) { reads1.push_back(read1); reads2.push_back(read2); }
            } else if (!jump_far) { reads1.push_back(read1); reads2.push_back(read2); }
        }
        if (!jump_far) print_vectors(S, reads1, reads2);
    }
    fclose(mapFile);
    if (S.out1) fclose(S.out1);
    if (S.out2) fclose(S.out2);
    // ---- results
    out.gaps = S.gaps; out.perfect_gap = S.perfect_gap; out.perfect_len = S.perfect_len;
    out.totalCount = S.totalCount; out.unCount = S.unCount; out.maxReadLength = S.maxReadLength;
    if (write_mode) {
        auto put = [&](const std::string &path, const std::string &text) { FILE *f = fopen(path.c_str(), "w"); if (!f) return false; fwrite(text.data(), 1, text.size(), f); fclose(f); return true; };
        if (!put(a.tmpDir + "gapInfo.txt", gapInfo)) { err = "can't write gapInfo.txt"; return 1; }
        if (fwrite(S.myout.data(), 1, S.myout.size(), S.myout_file) != S.myout.size() || S.myout_failed) { fclose(S.myout_file); err = "Can't create myout file"; return 1; }
        fclose(S.myout_file); S.myout_file = nullptr; S.myout.clear();
        char buf[128];
        snprintf(buf, sizeof buf, "%ld %ld %ld %ld", S.totalCount, S.unCount, (long)S.maxReadLength, kMaxFragment);
        if (!put(a.tmpDir + "stat.txt", buf)) { err = "can't write stat.txt"; return 1; }
        std::string st2;
        for (size_t g = 0; g < ng; g++) { snprintf(buf, sizeof buf, "%d\t%d\t%d\n", 1, S.perfect_gap[g], S.perfect_len[g]); st2 += buf; }
        if (!put(a.tmpDir + "stat2.txt", st2)) { err = "can't write stat2.txt"; return 1; }
        for (size_t g = 0; g < ng && write_mode == 1; g++) {
            const std::string nm = a.gapsDir + (a.samflag == 2 ? "gaps_" : "partial_gaps_") + std::to_string(g) + ".sam";
            if (!put(nm, a.samflag == 2 ? S.gap_text[g] : S.partial_text[g])) { err = "can't write " + nm; return 1; }
        }
    }
    out.gap_files = std::move(S.gap_text); out.partial_files = std::move(S.partial_text);
    return 0;
}

int preprocess_main(int argc, char **argv) {
    if (argc < 14) { fprintf(stderr, "Invalid parameters\n"); return 1; }
    Args a;
    a.contigFile = argv[1]; a.maxDistance = atoi(argv[2]); a.samflag = atoi(argv[3]); a.mapFile = argv[4]; a.outFile = argv[5];
    a.filledContigFile = argv[6]; a.reads1 = argv[7]; a.reads2 = argv[8]; a.gapsDir = argv[9]; a.tmpDir = argv[10];
    a.default_setting = atoi(argv[11]); a.genome_reduction = atoi(argv[12]); a.read_reduction = atoi(argv[13]);
    Binned B; std::string err;
    int rc = preprocess(a, B, err);
    for (const std::string &l : B.stdout_lines) printf("%s\n", l.c_str());
    if (rc) fprintf(stderr, "%s\n", err.c_str());
    return rc;
}

}  // namespace figsam
