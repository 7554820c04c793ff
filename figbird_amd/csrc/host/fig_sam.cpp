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
#include <chrono>
#include <string_view>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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
    std::vector<std::unordered_set<std::string>> partial_set;
    // index of the duplicate test's substring scan (dup_jump): per gap, every window of length jump_wlen[g] of every kept read,
    // keyed by its hash -> (read, offset); jump_wlen[g] = (length of the gap's first kept read) - 4, i.e. the core length of a
    // query of that same length
    struct Win { size_t h; uint32_t read, off; };
    std::vector<std::vector<Win>> jump_win;              // (a gap keeps a few dozen reads: a flat list scanned for the hash beats a node-based map)
    std::vector<long> jump_wlen;
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
    static const struct Tab { char c[256]; Tab() { memset(c, 'N', sizeof c); c[(unsigned char)'A'] = 'T'; c[(unsigned char)'C'] = 'G'; c[(unsigned char)'G'] = 'C'; c[(unsigned char)'T'] = 'A'; } } tab;
    std::string r(s.size(), 'N');
    const size_t n = s.size();
    for (size_t i = 0; i < n; i++) r[n - 1 - i] = tab.c[(unsigned char)s[i]];
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
    static const struct Tab { bool ok[256]; Tab() { memset(ok, 0, sizeof ok); for (const char *p = "ACGTNacgtn"; *p; p++) ok[(unsigned char)*p] = true; } } tab;
    for (char c : r) if (!tab.ok[(unsigned char)c]) return true;       // (a NUL inside the string counts as outside, as strchr(..., 0) != NULL never arises for std::string data)
    return false;
}
bool ncount_ok(const std::string &r) { int c = 0; for (char ch : r) if (ch == 'N') c++; return c <= 3; }      // check_Ncount_partial, :857-866

bool dup_jump(State &S, const std::string &read, int g) {      // check_duplicate, samflag 2 (:369-387)
    // (the reference first looks for an identical kept read, :369-375: an identical read also holds the clipped core as a
    //  substring, and with an empty core any kept read makes the test true, so the scan below decides both)
    if (S.read_count[g] == 0) return false;
    const std::string core = read.size() >= 4 ? read.substr(2, read.size() - 4) : std::string();       // clip 2 from either end
    // (memmem: glibc's vectorised two-way search; this scan over every read already kept for the gap is the reference's own
    //  quadratic duplicate test and dominates the ingest at thousands of reads per gap)
    if (core.empty()) return !S.jump_reads[g].empty();
    if ((long)core.size() == S.jump_wlen[g]) {
        // every kept read that is long enough has all its windows of this length in the index: a hash hit verified by memcmp
        // is "core is a substring of a kept read", and no hit means none is
        const size_t h = std::hash<std::string_view>()(std::string_view(core));
        for (const State::Win &w : S.jump_win[g])
            if (w.h == h && memcmp(S.jump_reads[g][w.read].data() + w.off, core.data(), core.size()) == 0) return true;
        return false;
    }
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

void store_jump(State &S, int g, const std::string &seq) {
    if (S.jump_reads[g].empty()) S.jump_wlen[g] = seq.size() > 4 ? (long)seq.size() - 4 : -1;
    const uint32_t idx = (uint32_t)S.jump_reads[g].size();
    S.jump_reads[g].push_back(seq); S.read_count[g]++;
    const long w = S.jump_wlen[g];
    if (w > 0 && (long)seq.size() >= w)
        for (uint32_t o = 0; o + (size_t)w <= seq.size(); o++) S.jump_win[g].push_back({std::hash<std::string_view>()(std::string_view(seq.data() + o, (size_t)w)), idx, o});
}

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


// =====================================================================================================================
// The fast form of the stage (round 4).  Same decisions, same bytes; what changes is how the records travel:
//   * the SAM is mapped once and parsed once, in parallel chunks, into 48-byte records of offsets into the mapping (the
//     reference -- and the legacy form above -- tokenises every line into seven heap strings, twice for a far jump library);
//   * the two passes of a far jump library (:2278-2435 insert-size mean, :2437-2566 binning) and the pair-grouping logic walk
//     that array; records that go to myout.sam are only NOTED (record, IH) and the insert-size histogram is fed from the
//     parsed fields (what processMapping would re-read from the line, :1768-1830);
//   * myout.sam is formatted from the notes in parallel chunks and written with large writes.
// The legacy form stays for inputs whose quirks the fast form does not restate: a line of 1023 bytes or more (the reference's
// 1024-byte fgets splits it), a header line behind the first record, a record with fewer than 11 fields, the reduced-read
// files (rewrite_readset reverses qualities in place and later writers see it), or FIGSAM_LEGACY=1 (A/B tests).
// =====================================================================================================================
typedef std::string_view sv;

struct RecC {
    uint64_t off;                                    // start of the line in the mapping
    uint16_t qn_o, qn_l, rn_o, rn_l, cg_o, cg_l, sq_o, sq_l, ql_o, ql_l, md_o, md_l;
    int32_t flag, pos, tlen, contigNo;
};

struct SamV {                                        // a record as the pair logic sees it: views into the mapping
    sv qname, rname, cigar, seq, qual, md; int flag = 0, pos = 0, tlen = 0; long ih = 1, contigNo = -1;
};

inline int sv_atoi(const char *p, const char *e) {   // atoi on a token: optional sign, digits, stops at the first other character
    long v = 0; bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
    return (int)(neg ? -v : v);
}

struct NameMap {                                     // getContigNo (:327-338): first contig of that name wins
    std::vector<int> slot; std::vector<sv> key; std::vector<long> val; size_t mask = 0;
    static uint64_t h(sv s) { uint64_t x = 1469598103934665603ull; for (char c : s) { x ^= (unsigned char)c; x *= 1099511628211ull; } return x; }
    void build(const std::vector<std::string> &names, size_t n) {
        size_t cap = 16; while (cap < 2 * n + 8) cap <<= 1;
        mask = cap - 1; slot.assign(cap, -1);
        for (size_t i = 0; i < n; i++) {
            sv k(names[i]); size_t q = h(k) & mask; bool dup = false;
            while (slot[q] >= 0) { if (key[(size_t)slot[q]] == k) { dup = true; break; } q = (q + 1) & mask; }
            if (!dup) { slot[q] = (int)key.size(); key.push_back(k); val.push_back((long)i); }
        }
    }
    long find(sv k) const {
        if (slot.empty()) return -1;
        size_t q = h(k) & mask;
        while (slot[q] >= 0) { if (key[(size_t)slot[q]] == k) return val[(size_t)slot[q]]; q = (q + 1) & mask; }
        return -1;
    }
};

// one chunk of lines [b, e) of the mapping -> records; false when a line needs the legacy form
bool parse_chunk(const char *base, size_t b, size_t e, const NameMap &names, std::vector<RecC> &out, bool &saw_record, bool &has_header, bool &header_after_record) {
    sv last_name; long last_no = -1; bool have_last = false;
    size_t p = b;
    while (p < e) {
        const char *ls = base + p;
        const char *nl = (const char *)memchr(ls, '\n', e - p);
        const size_t len = nl ? (size_t)(nl - ls) : e - p;
        if (len + (nl ? 1 : 0) >= (size_t)kRec - 1) return false;          // fgets(line, 1024) would split it
        p += len + (nl ? 1 : 0);
        if (len == 0) return false;                                          // an empty line: get_sam fails on it (legacy decides)
        if (ls[0] == '@') { has_header = true; if (saw_record) header_after_record = true; continue; }
        saw_record = true;
        RecC r; memset(&r, 0, sizeof r); r.off = (uint64_t)(ls - base);
        const char *q = ls, *le = ls + len;
        const char *tok[11]; const char *tend[11]; int nt = 0;
        const char *mdp = nullptr, *mde = nullptr;
        while (q < le) {
            while (q < le && (*q == '\t' || *q == ' ')) q++;
            if (q >= le) break;
            const char *t0 = q;
            while (q < le && *q != '\t' && *q != ' ') q++;
            if (nt < 11) { tok[nt] = t0; tend[nt] = q; nt++; }
            else if (q - t0 >= 2 && t0[0] == 'M' && t0[1] == 'D') { mdp = t0; mde = q; }
        }
        if (nt < 11) return false;
        auto set = [&](uint16_t &o, uint16_t &l, int k) { o = (uint16_t)(tok[k] - ls); l = (uint16_t)(tend[k] - tok[k]); };
        set(r.qn_o, r.qn_l, 0); set(r.rn_o, r.rn_l, 2); set(r.cg_o, r.cg_l, 5); set(r.sq_o, r.sq_l, 9); set(r.ql_o, r.ql_l, 10);
        if (mdp) { r.md_o = (uint16_t)(mdp - ls); r.md_l = (uint16_t)(mde - mdp); }
        r.flag = sv_atoi(tok[1], tend[1]); r.pos = sv_atoi(tok[3], tend[3]); r.tlen = sv_atoi(tok[8], tend[8]);
        const sv rn(tok[2], (size_t)(tend[2] - tok[2]));
        if (!have_last || rn != last_name) { last_no = names.find(rn); last_name = rn; have_last = true; }
        r.contigNo = (int32_t)last_no;
        out.push_back(r);
    }
    return true;
}

struct Fast {
    State &S; const char *base; const std::vector<RecC> &R;
    std::vector<std::pair<uint32_t, uint32_t>> emit;       // myout.sam: (record, IH) in file order
    Fast(State &s, const char *b, const std::vector<RecC> &r) : S(s), base(b), R(r) {}
    SamV view(size_t i) const {
        const RecC &r = R[i]; const char *l = base + r.off; SamV v;
        v.qname = sv(l + r.qn_o, r.qn_l); v.rname = sv(l + r.rn_o, r.rn_l); v.cigar = sv(l + r.cg_o, r.cg_l); v.seq = sv(l + r.sq_o, r.sq_l);
        v.qual = sv(l + r.ql_o, r.ql_l); v.md = sv(l + r.md_o, r.md_l); v.flag = r.flag; v.pos = r.pos; v.tlen = r.tlen; v.contigNo = r.contigNo;
        return v;
    }
};

char *put_int(char *p, long v);
void sam_line_v(std::string &dst, const SamV &r, sv seq) {     // writeSam / writeSam2 (:404-417) with the sequence as it stands
    const size_t at = dst.size();
    dst.resize(at + r.qname.size() + r.cigar.size() + seq.size() + r.qual.size() + r.md.size() + 96);
    char *w = &dst[at];
    memcpy(w, r.qname.data(), r.qname.size()); w += r.qname.size(); *w++ = '\t';
    w = put_int(w, r.flag); *w++ = '\t'; w = put_int(w, r.contigNo); *w++ = '\t'; w = put_int(w, r.pos); *w++ = '\t';
    memcpy(w, r.cigar.data(), r.cigar.size()); w += r.cigar.size(); *w++ = '\t';
    w = put_int(w, r.tlen); *w++ = '\t';
    memcpy(w, seq.data(), seq.size()); w += seq.size(); *w++ = '\t';
    memcpy(w, r.qual.data(), r.qual.size()); w += r.qual.size(); *w++ = '\t';
    memcpy(w, r.md.data(), r.md.size()); w += r.md.size();
    memcpy(w, "\tIH:i:", 6); w += 6; w = put_int(w, r.ih); *w++ = '\n';
    dst.resize((size_t)(w - &dst[0]));
}

char *put_int(char *p, long v) {                     // decimal, as %ld prints it
    char t[24]; int n = 0; unsigned long u = v < 0 ? (unsigned long)(-(v + 1)) + 1ul : (unsigned long)v;
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = t[--n];
    return p;
}

double n_frac_v(sv s) { int c = 0; for (char ch : s) if (ch == 'N' || ch == 'n') c++; return c / (double)s.size(); }

// what account_myout_line reads back from the line myout_line has just written (processMapping, :1768-1830)
void account_fields(State &S, const SamV &r) {
    const long nh = r.ih;
    // (the re-read scans the tokens behind the sequence for "MD" / "IH" prefixes, the quality string included: a quality string
    //  that starts with "MD" stands in for a missing MD tag; the real IH token comes last and wins)
    const sv md = !r.md.empty() ? r.md : ((r.qual.size() >= 2 && r.qual[0] == 'M' && r.qual[1] == 'D') ? r.qual : sv());
    if (nh == 1 && !(md.size() > 5 && md[5] == '^')) {
        const long c = r.contigNo;
        if (c >= 0 && c < (long)S.contigs.size() && !S.contigs[(size_t)c].empty()) {
            const int insertSize = r.tlen;
            if (insertSize > 0) { if (insertSize < kMaxFragment) S.insertCounts[(size_t)insertSize]++; else if (insertSize > kMaxFragment) S.discarded++; }
        }
    }
}

void print_vectors_v(Fast &F, std::vector<uint32_t> &reads1, std::vector<uint32_t> &reads2) {     // printVectors, :641-855
    State &S = F.S;
    const unsigned long ih = reads1.size();
    for (unsigned long i = 0; i < ih; i++) {
        SamV r1 = F.view(reads1[i]), r2 = F.view(reads2[i]);
        if (r1.rname == "*" || r2.rname == "*") {
            if (r1.seq.size() > S.maxReadLength) S.maxReadLength = r1.seq.size();
            if (r2.seq.size() > S.maxReadLength) S.maxReadLength = r2.seq.size();
            if (n_frac_v(r1.seq) < 0.8 && n_frac_v(r2.seq) < 0.8) { S.unCount++; S.totalCount++; reads1.clear(); reads2.clear(); return; }
        } else if (r1.rname != r2.rname) {
        } else {
            if (r1.seq.size() > S.maxReadLength) S.maxReadLength = r1.seq.size();
            if (r2.seq.size() > S.maxReadLength) S.maxReadLength = r2.seq.size();
            r1.ih = (long)ih; r2.ih = (long)ih;
            F.emit.emplace_back(reads1[i], (uint32_t)ih); F.emit.emplace_back(reads2[i], (uint32_t)ih);
            if (S.account) { account_fields(S, r1); account_fields(S, r2); }
        }
    }
    S.totalCount++;
    reads1.clear(); reads2.clear();
}

void collect_partial_v(State &S, const SamV &read, int pos2) {          // collectPartialSAM :1667-1694
    const int strandNo = (read.flag & 16) >> 4;
    const std::string cigar(read.cigar);
    const int del = parse_del(cigar);
    const int g = check_pos2(S, read.contigNo, read.pos, (int)read.seq.size(), del);
    if (g >= 0 && S.partial_read_count[g] <= kReadCap) {
        const std::string seq(read.seq);
        if (ncount_ok(seq) && !S.partial_set[g].count(seq)) {
            Sam tmp; tmp.seq = seq; tmp.cigar = cigar; tmp.pos = read.pos; tmp.qual = std::string(read.qual);
            write_partial(S, tmp, g, strandNo, del, pos2);
            S.partial_set[g].insert(seq);
            S.partial_read_count[g]++;
            check_mim(S, cigar, g);
        }
    }
}

void print_mixed_v(Fast &F, std::vector<uint32_t> &x1, std::vector<uint32_t> &x2) {              // printMixedVectors, :999-1489
    State &S = F.S;
    const int maxD = S.a.maxDistance, samflag = S.a.samflag;
    for (size_t oi = 0; oi < x1.size(); oi++) {
        for (size_t oj = 0; oj < x2.size(); oj++) {
            const SamV read1 = F.view(x1[oi]), read2 = F.view(x2[oj]);
            if (oi == 0 && oj == 0) {
                if (read1.seq.size() > S.maxReadLength) S.maxReadLength = read1.seq.size();
                if (read2.seq.size() > S.maxReadLength) S.maxReadLength = read2.seq.size();
                if (n_frac_v(read1.seq) < 0.8 && n_frac_v(read2.seq) < 0.8) { S.unCount++; S.totalCount++; }
                else { x1.clear(); x2.clear(); return; }
            }
            if ((read1.flag & 4) != 0 && (read2.flag & 4) != 0) { x1.clear(); x2.clear(); return; }
            if (((read1.flag & 4) == 0 && (read2.flag & 4) != 0) || ((read1.flag & 4) == 0 && (read2.flag & 4) == 0 && maxD > 250)) {
                const bool r2_unmapped = (read2.flag & 4) != 0;
                const SamV m2 = F.view(x2[0]);
                std::string m2seq(m2.seq);                              // m2[0].seq is rewritten as mates are stored (:1236, :1290)
                for (size_t i = 0; i < x1.size(); i++) {
                    const SamV m1 = F.view(x1[i]);
                    std::string m1seq(m1.seq);
                    const long contigNo1 = m1.contigNo; const long pos1 = m1.pos; const int strandNo1 = (m1.flag & 16) >> 4;
                    if (samflag == 2 && !check_char(m2seq)) {
                        if (r2_unmapped) {
                            const int g = check_pos(S, contigNo1, pos1, strandNo1, (int)m2seq.size());
                            if (g >= 0 && S.read_count[g] <= kReadCap) {
                                const std::string temp = revcomp(m2seq);
                                if ((strandNo1 == 1 && !dup_jump(S, m2seq, g)) || (strandNo1 == 0 && !dup_jump(S, temp, g))) {
                                    sam_line_v(S.gap_text[g], m1, m1seq); sam_line_v(S.gap_text[g], m2, m2seq);
                                    if (strandNo1 == 0) m2seq = temp;
                                    store_jump(S, g, m2seq);
                                }
                            }
                        } else {
                            const std::string original2 = m2seq, original1 = m1seq;
                            const long contigNo2 = m2.contigNo; const long pos2 = m2.pos; const int strandNo2 = (m2.flag & 16) >> 4;
                            int g = check_pos(S, contigNo1, pos1, strandNo1, (int)m2seq.size());
                            if (g >= 0 && S.read_count[g] <= kReadCap) {
                                const std::string temp = revcomp(m2seq);
                                if (strandNo2 == 1) m2seq = temp;
                                const std::string rev = revcomp(m2seq);
                                if ((strandNo1 == 1 && !dup_jump(S, m2seq, g)) || (strandNo1 == 0 && !dup_jump(S, rev, g))) {
                                    sam_line_v(S.gap_text[g], m1, m1seq); sam_line_v(S.gap_text[g], m2, m2seq);
                                    if (strandNo1 == 0) m2seq = rev;
                                    store_jump(S, g, m2seq);
                                }
                            }
                            m2seq = original2; m1seq = original1;
                            g = check_pos(S, contigNo2, pos2, strandNo2, (int)m1seq.size());
                            if (g >= 0 && S.read_count[g] <= kReadCap) {
                                const std::string temp = revcomp(m1seq);
                                if (strandNo1 == 1) m1seq = temp;
                                const std::string rev = revcomp(m1seq);
                                if ((strandNo2 == 1 && !dup_jump(S, m1seq, g)) || (strandNo2 == 0 && !dup_jump(S, rev, g))) {
                                    sam_line_v(S.gap_text[g], m2, m2seq); sam_line_v(S.gap_text[g], m1, m1seq);
                                    if (strandNo2 == 0) m1seq = rev;
                                    store_jump(S, g, m1seq);
                                }
                            }
                            // (m1[i].seq keeps what the last branch left in it; nothing reads it again: the vectors are cleared below)
                        }
                    }
                    if (samflag == 1) collect_partial_v(S, m1, -1);
                }
                x1.clear(); x2.clear();
                return;
            }
        }
    }
    x1.clear(); x2.clear();
}

// ---- the whole stage on the parsed records; returns -1 when the input needs the legacy form
int preprocess_fast(State &S, const Args &a, FILE *myout_file, bool &myout_failed) {
    const bool jump_far = a.samflag == 2 && a.maxDistance > 250;
    const bool timing = getenv("FIGSAM_TIMING") != nullptr;                    // stage timers on stderr (tools/time_host_stages.py)
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = tnow();
    auto lap = [&](const char *what) { if (timing) { const double t = tnow(); fprintf(stderr, "[figsam] %-28s %.3f s\n", what, t - t_last); t_last = t; } };
    int fd = open(a.mapFile.c_str(), O_RDONLY);
    if (fd < 0) return -1;
    struct stat stt;
    if (fstat(fd, &stt) != 0 || stt.st_size == 0) { close(fd); return -1; }
    const size_t fsz = (size_t)stt.st_size;
    const char *base = (const char *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return -1;
    NameMap names; names.build(S.contigNames, std::min(S.contigs.size(), S.contigNames.size()));
    // ---- parallel parse
    int nthr = (int)std::thread::hardware_concurrency(); if (nthr < 1) nthr = 1; if (nthr > 16) nthr = 16;
    if (const char *ev = getenv("FIGFILL_THREADS")) { int v = atoi(ev); if (v >= 1 && v <= 64) nthr = v; }
    if (fsz < (1u << 16)) nthr = 1;
    std::vector<size_t> cut((size_t)nthr + 1, fsz); cut[0] = 0;
    for (int t = 1; t < nthr; t++) { size_t p = fsz / (size_t)nthr * (size_t)t; const char *nl = (const char *)memchr(base + p, '\n', fsz - p); cut[(size_t)t] = nl ? (size_t)(nl - base) + 1 : fsz; }
    std::vector<std::vector<RecC>> parts((size_t)nthr); std::vector<char> okv((size_t)nthr, 1), sawv((size_t)nthr, 0), hasv((size_t)nthr, 0), hdrv((size_t)nthr, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; t++) th.emplace_back([&, t] {
            bool saw = false, has = false, hdr = false;
            parts[(size_t)t].reserve((cut[(size_t)t + 1] - cut[(size_t)t]) / 200 + 16);
            okv[(size_t)t] = parse_chunk(base, cut[(size_t)t], cut[(size_t)t + 1], names, parts[(size_t)t], saw, has, hdr) ? 1 : 0;
            sawv[(size_t)t] = saw; hasv[(size_t)t] = has; hdrv[(size_t)t] = hdr; });
        for (auto &t : th) t.join();
    }
    lap("map + parallel parse");
    bool ok = true, saw = false;                       // header lines only in front of the first record (the loops skip '@' lines at their top level only)
    for (int t = 0; t < nthr && ok; t++) {
        if (!okv[(size_t)t] || hdrv[(size_t)t] || (saw && hasv[(size_t)t])) ok = false;
        if (sawv[(size_t)t]) saw = true;
    }
    if (!ok) { munmap((void *)base, fsz); return -1; }
    std::vector<RecC> R;
    { size_t n = 0; for (auto &v : parts) n += v.size(); R.reserve(n); for (auto &v : parts) { R.insert(R.end(), v.begin(), v.end()); std::vector<RecC>().swap(v); } }
    if (R.size() >= 0xffffffffull) { munmap((void *)base, fsz); return -1; }
    lap("concatenate records");
    Fast F(S, base, R);
    const size_t NR = R.size();
    auto qn = [&](size_t i) { const RecC &r = R[i]; return sv(base + r.off + r.qn_o, r.qn_l); };
    std::vector<uint32_t> reads1, reads2, mixed1, mixed2;
    sv preq1 = "*", preq2 = "*";
    if (jump_far) {
        // ---- first pass (:2278-2435)
        S.account = true; S.insertCounts.assign((size_t)kMaxFragment, 1); S.discarded = 0;
        size_t i = 0; bool end = false;
        while (i < NR) {
            size_t r1 = i++;
            while ((R[r1].flag & 2) == 0) {
                for (int seg_no = 0; seg_no < 2 && !end; seg_no++) {
                    const sv q = qn(r1); const int seg = R[r1].flag & 192;
                    while (q == qn(r1) && (R[r1].flag & 192) == seg) { if (i >= NR) { end = true; break; } r1 = i++; }
                }
                if (end) break;
            }
            if (end) break;
            if (i >= NR) break;
            const size_t r2 = i++;
            if (qn(r1) != preq1 || qn(r2) != preq2) { preq1 = qn(r1); preq2 = qn(r2); print_vectors_v(F, reads1, reads2); }
            reads1.push_back((uint32_t)r1); reads2.push_back((uint32_t)r2);
        }
        print_vectors_v(F, reads1, reads2);
        S.account = false;
        long insCount = S.discarded; double sum = 0;
        for (long k = 0; k < kMaxFragment; k++) { insCount += S.insertCounts[(size_t)k] - 1; sum += k * (S.insertCounts[(size_t)k] - 1); }
        S.read_mean = (int)(sum / insCount);
        preq1 = "*"; preq2 = "*";
        lap("first pass (insert mean)");
    }
    // ---- main pass (:2437-2566)
    {
        size_t i = 0; bool end = false;
        while (i < NR) {
            size_t r1 = i++;
            while ((R[r1].flag & 2) == 0) {
                {   const sv q = qn(r1); const int seg = R[r1].flag & 192;
                    while (q == qn(r1) && (R[r1].flag & 192) == seg) { mixed1.push_back((uint32_t)r1); if (i >= NR) { end = true; break; } r1 = i++; } }
                if (end) { mixed1.clear(); break; }
                {   const sv q = qn(r1); const int seg = R[r1].flag & 192;
                    while (q == qn(r1) && (R[r1].flag & 192) == seg) { mixed2.push_back((uint32_t)r1); if (i >= NR) { end = true; break; } r1 = i++; } }
                print_mixed_v(F, mixed1, mixed2);
                if (end) break;
            }
            if (end) break;
            if (i >= NR) break;
            const size_t r2 = i++;
            if (qn(r1) != preq1 || qn(r2) != preq2) {
                preq1 = qn(r1); preq2 = qn(r2);
                if (!jump_far) print_vectors_v(F, reads1, reads2);
                if (a.samflag == 1) {
                    const SamV v1 = F.view(r1), v2 = F.view(r2);
                    const bool f1 = v1.cigar == "101M", f2 = v2.cigar == "101M";
                    if (!(f1 && f2)) { collect_partial_v(S, v1, v2.pos); collect_partial_v(S, v2, v1.pos); }
                }
                if (!jump_far) { reads1.push_back((uint32_t)r1); reads2.push_back((uint32_t)r2); }
            } else if (!jump_far) { reads1.push_back((uint32_t)r1); reads2.push_back((uint32_t)r2); }
        }
        if (!jump_far) print_vectors_v(F, reads1, reads2);
    }
    lap("main pass (binning)");
    // ---- myout.sam from the notes.  The sizes of the lines are known from the parsed fields, so every block of records has its
    // place in the file before a byte is formatted: the file is sized once, mapped, and the blocks are formatted straight into
    // the mapping by all threads (a single stream of write() calls copies ~0.3 GB/s into the page cache on this class of host;
    // page faults on a shared mapping scale with the threads).  Falls back to buffered writes when the mapping fails.
    if (myout_file) {
        const size_t NE = F.emit.size();
        const size_t per = 1u << 15;                                   // records per block
        const size_t nblk = (NE + per - 1) / per;
        auto ndig = [](long v) { size_t n = v < 0 ? 2 : 1; unsigned long u = v < 0 ? (unsigned long)(-(v + 1)) + 1ul : (unsigned long)v; while (u >= 10) { u /= 10; n++; } return n; };
        auto fmt = [&](size_t e0, size_t e1, char *w) {
            for (size_t e = e0; e < e1; e++) {
                const RecC &r = R[F.emit[e].first]; const char *l = base + r.off;
                memcpy(w, l + r.qn_o, r.qn_l); w += r.qn_l; *w++ = '\t';
                w = put_int(w, r.flag); *w++ = '\t'; w = put_int(w, r.contigNo); *w++ = '\t'; w = put_int(w, r.pos); *w++ = '\t';
                memcpy(w, l + r.cg_o, r.cg_l); w += r.cg_l; *w++ = '\t';
                w = put_int(w, r.tlen); *w++ = '\t';
                memcpy(w, l + r.sq_o, r.sq_l); w += r.sq_l; *w++ = '\t';
                memcpy(w, l + r.ql_o, r.ql_l); w += r.ql_l; *w++ = '\t';
                memcpy(w, l + r.md_o, r.md_l); w += r.md_l;
                memcpy(w, "\tIH:i:", 6); w += 6; w = put_int(w, (long)F.emit[e].second); *w++ = '\n';
            }
            return w;
        };
        std::vector<size_t> boff(nblk + 1, 0);
        {   // sizes of the blocks (parallel), then their offsets
            std::vector<std::thread> th;
            for (int t = 0; t < nthr; t++) th.emplace_back([&, t] {
                for (size_t b = (size_t)t; b < nblk; b += (size_t)nthr) {
                    const size_t e0 = b * per, e1 = std::min(NE, e0 + per);
                    size_t n = 0;
                    for (size_t e = e0; e < e1; e++) {
                        const RecC &r = R[F.emit[e].first];
                        n += (size_t)r.qn_l + r.cg_l + r.sq_l + r.ql_l + r.md_l + ndig(r.flag) + ndig(r.contigNo) + ndig(r.pos) + ndig(r.tlen) + ndig((long)F.emit[e].second) + 15;
                    }
                    boff[b + 1] = n;
                } });
            for (auto &t : th) t.join();
            for (size_t b = 0; b < nblk; b++) boff[b + 1] += boff[b];
        }
        const size_t total = boff[nblk];
        bool mapped = false;
        fflush(myout_file);
        const int ofd = fileno(myout_file);
        if (total > 0 && ftell(myout_file) == 0 && ofd >= 0 && ftruncate(ofd, (off_t)total) == 0) {
            // (fopen(..., "w") gave a write-only descriptor: a shared mapping needs read access as well)
            int rfd = open(a.outFile.c_str(), O_RDWR);
            char *om = rfd >= 0 ? (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, rfd, 0) : (char *)MAP_FAILED;
            if (rfd >= 0) close(rfd);
            if (om != (char *)MAP_FAILED) {
                std::vector<std::thread> th;
                std::vector<char> bad((size_t)nthr, 0);
                for (int t = 0; t < nthr; t++) th.emplace_back([&, t] {
                    for (size_t b = (size_t)t; b < nblk; b += (size_t)nthr) {
                        const size_t e0 = b * per, e1 = std::min(NE, e0 + per);
                        char *w = fmt(e0, e1, om + boff[b]);
                        if ((size_t)(w - om) != boff[b + 1]) bad[(size_t)t] = 1;
                    } });
                for (auto &t : th) t.join();
                for (char c : bad) if (c) myout_failed = true;        // (a size formula out of step with the formatter: never observed)
                munmap(om, total);
                fseek(myout_file, 0, SEEK_END);
                mapped = true;
            } else if (ftruncate(ofd, 0) != 0) myout_failed = true;
        }
        if (!mapped) {
            std::vector<char> buf;
            for (size_t b = 0; b < nblk; b++) {
                buf.resize(boff[b + 1] - boff[b]);
                fmt(b * per, std::min(NE, b * per + per), buf.data());
                if (fwrite(buf.data(), 1, buf.size(), myout_file) != buf.size()) myout_failed = true;
            }
        }
    }
    lap("myout.sam format + write");
    munmap((void *)base, fsz);
    return 0;
}

}  // namespace

int preprocess(const Args &a, Binned &out, std::string &err, int write_mode) {
    State S; S.a = a;
    const bool timing = getenv("FIGSAM_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = tnow();
    auto lap = [&](const char *what) { if (timing) { const double t = tnow(); fprintf(stderr, "[figsam] %-28s %.3f s\n", what, t - t_last); t_last = t; } };
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
    lap("scaffold + gaps");
    const size_t ng = S.gaps.size();
    S.read_count.assign(ng, 0); S.partial_read_count.assign(ng, 0); S.jump_reads.assign(ng, {}); S.partial_set.assign(ng, {});
    S.jump_win.assign(ng, {}); S.jump_wlen.assign(ng, -1);
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
    // the fast form (mapped file, records parsed once in parallel, myout.sam formatted in parallel) unless the input needs the
    // legacy one below; both produce the same bytes
    bool fast_done = false;
    if (!S.writeflag && !getenv("FIGSAM_LEGACY")) {
        const int frc = preprocess_fast(S, a, S.myout_file, S.myout_failed);
        if (frc == 0) fast_done = true;
        else {      // nothing has been written or counted yet unless the parse succeeded: start the legacy form from a clean state
            S.totalCount = S.unCount = 0; S.maxReadLength = 0; S.read_mean = 0; S.account = false;
        }
    }
    if (fast_done) { fclose(mapFile); mapFile = nullptr; }
    else if (jump_far) {
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
    if (!fast_done) {
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
    if (mapFile) fclose(mapFile);
    if (S.out1) fclose(S.out1);
    if (S.out2) fclose(S.out2);
    lap("SAM passes (total)");
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
        if (write_mode == 1) {
            // one file per gap (the reference re-opens the gap's file for every read it bins, :404-410): creating ~10^4 files is
            // what takes the time, so the files are dealt over a few threads
            int nth = (int)std::thread::hardware_concurrency(); if (nth < 1) nth = 1; if (nth > 8) nth = 8;
            if (ng < 64) nth = 1;
            std::vector<long> bad((size_t)nth, -1);
            std::vector<std::thread> th;
            for (int t = 0; t < nth; t++) th.emplace_back([&, t] {
                for (size_t g = (size_t)t; g < ng; g += (size_t)nth) {
                    const std::string nm = a.gapsDir + (a.samflag == 2 ? "gaps_" : "partial_gaps_") + std::to_string(g) + ".sam";
                    if (!put(nm, a.samflag == 2 ? S.gap_text[g] : S.partial_text[g]) && bad[(size_t)t] < 0) bad[(size_t)t] = (long)g;
                } });
            for (auto &t : th) t.join();
            for (long g : bad) if (g >= 0) { err = "can't write " + a.gapsDir + (a.samflag == 2 ? "gaps_" : "partial_gaps_") + std::to_string(g) + ".sam"; return 1; }
        }
    }
    out.gap_files = std::move(S.gap_text); out.partial_files = std::move(S.partial_text);
    lap("run-level + per-gap files");
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
