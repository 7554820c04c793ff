// fig_host.cpp -- see fig_host.h.  Reference citations are to /root/reference (Figbird.cpp,
// FillGaps.cpp, Preprocess.cpp); behaviour (including quirks) is kept.
#include "fig_host.h"
#include "../fig_gaprules.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace fighost {

static const int kMaxRec = 1024;            // MAX_REC_LEN: every reader uses fgets(line, 1024)
static const int kReadCap = 3000;           // partial_limit / unmapped_limit

// ------------------------------------------------------------------ scaffold
bool load_scaffold(const std::string &path, Scaffold &sc, std::string &err) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) { err = "Can't open contig file"; return false; }
    sc.names.clear(); sc.off.clear(); sc.seq.clear();
    sc.off.push_back(0);
    char *line = nullptr; size_t cap = 0; ssize_t n;
    int64_t curlen = 0;
    while ((n = getline(&line, &cap, f)) != -1) {
        if (line[0] == ';') continue;
        if (line[0] == '>') {
            // name = first whitespace-delimited token of the header (Figbird.cpp:6994-6997)
            std::string h(line + 1, (size_t)n - 1);
            if (!h.empty()) h.pop_back();
            size_t b = h.find_first_not_of(" \t\n");
            std::string tok;
            if (b != std::string::npos) { size_t e = h.find_first_of(" \t\n", b); tok = h.substr(b, e == std::string::npos ? std::string::npos : e - b); }
            sc.names.push_back(tok);
            if (curlen > 0) { sc.off.push_back((int64_t)sc.seq.size()); curlen = 0; }
        } else {
            // every sequence line loses its last character ('\n', or a base if the file does not end in one; :7030,7037)
            size_t len = (size_t)n;
            if (len > 0) len--;
            for (size_t i = 0; i < len; i++) sc.seq.push_back((char)toupper((unsigned char)line[i]));
            curlen += (int64_t)len;
        }
    }
    free(line);
    fclose(f);
    sc.off.push_back((int64_t)sc.seq.size());
    // a header with no sequence before the next header does not open a new contig in the reference; keep names aligned
    while ((int64_t)sc.names.size() < (int64_t)sc.off.size() - 1) sc.names.push_back("");
    return true;
}

// ------------------------------------------------------------------ model (A0)
namespace {

struct SamCols { char *qname, *cigar, *seq, *rname; int flag, pos, tlen; char md[1000]; int nh; bool ok; };

// token order of processMapping / computeLikelihood (Figbird.cpp:864-901, :1198-1227)
void split_sam(char *line, SamCols &c) {
    c.ok = false; c.md[0] = 0; c.nh = 0;
    char *sv = nullptr, *t;
    if (!(c.qname = strtok_r(line, "\t", &sv))) return;
    if (!(t = strtok_r(nullptr, "\t", &sv))) return; c.flag = atoi(t);
    if (!(c.rname = strtok_r(nullptr, "\t", &sv))) return;
    if (!(t = strtok_r(nullptr, "\t", &sv))) return; c.pos = atoi(t);
    if (!(c.cigar = strtok_r(nullptr, "\t", &sv))) return;
    if (!(t = strtok_r(nullptr, "\t", &sv))) return; c.tlen = atoi(t);
    if (!(c.seq = strtok_r(nullptr, "\t", &sv))) return;
    while ((t = strtok_r(nullptr, "\t\n", &sv)) != nullptr) {
        if (t[0] == 'M' && t[1] == 'D') { strncpy(c.md, t, sizeof(c.md) - 1); c.md[sizeof(c.md) - 1] = 0; }
        else if (t[0] == 'I' && t[1] == 'H') c.nh = atoi(t + 5);
    }
    c.ok = true;
}

inline int b5(char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }

struct Counts {
    int L = 0; int maxInsertSize = 0, MAX_INSERT_SIZE = 0;
    std::vector<long> insertCounts, errorPos, inPos, inLengths, delPos, delLengths, readLengths;
    long errorTypes[5][5]; long baseCounts[5];
    long discardedReads = 0, uniqueMappedReads = 0;
};

// walks a CIGAR the way the reference does (numbers split on "IDMS^", operator looked up by offset)
template <class F> void walk_cigar(const char *cigar, const char *delims, F f) {
    std::vector<char> tc(cigar, cigar + strlen(cigar) + 1);
    int totalLength = 0;
    char *svc = nullptr;
    for (char *t = strtok_r(tc.data(), delims, &svc); t; t = strtok_r(nullptr, delims, &svc)) {
        unsigned long n = (unsigned long)atoi(t);
        totalLength += (int)strlen(t);
        f(cigar[totalLength], n);
        totalLength++;
    }
}

// walks the mismatches of an MD tag (Figbird.cpp:378-484): calls f(index_in_read_1based_after_inserts..)
template <class F> void walk_md(const char *md, const std::vector<int> &inserts, const char *read, F f) {
    unsigned long mdLength = strlen(md) - 5;
    std::vector<char> tm(md, md + strlen(md) + 1);
    char *svm = nullptr;
    strtok_r(tm.data(), ":", &svm);
    strtok_r(nullptr, ":", &svm);
    int index = 0, totalLength = 0;
    char *temp;
    while ((temp = strtok_r(nullptr, "ACGTN^\t\n ", &svm)) != nullptr) {
        totalLength += (int)strlen(temp);
        if ((unsigned long)totalLength < mdLength) {
            char from = md[5 + totalLength];
            if (from == '^') {
                totalLength++;
                index += atoi(temp);
                for (unsigned long i = totalLength; i < mdLength; i++) {
                    from = md[5 + totalLength];
                    if (from == 'A' || from == 'C' || from == 'G' || from == 'T' || from == 'N') totalLength++; else break;
                }
            } else if (from == 'A' || from == 'C' || from == 'G' || from == 'T' || from == 'N') {
                totalLength++;
                index += atoi(temp) + 1;
                int curIndex = 0;
                for (int i = 0; i < index; i++) curIndex += inserts[i];
                f(index, curIndex, from, read[index - 1 + curIndex]);
            } else break;
        }
    }
}

void update_insert_counts(Counts &c, int index) {       // Figbird.cpp:186-225
    if (index <= 0) return;
    if (index < c.maxInsertSize) { c.insertCounts[index]++; return; }
    if (index > c.MAX_INSERT_SIZE) { c.discardedReads++; return; }
    int t = std::max(c.maxInsertSize * 2, index);
    c.insertCounts.resize((size_t)t + 1, 1);
    c.insertCounts[index]++;
    c.maxInsertSize = t;
}

void count_errors(Counts &c, const SamCols &s, const std::string &noErrorCigar) {   // processErrorTypes, :291-487
    int strandNo = (s.flag & 16) >> 4;
    const char *read = s.seq;
    int readLength = 0;
    for (; read[readLength]; readLength++) c.baseCounts[b5(read[readLength])]++;
    c.readLengths[readLength - 1]++;
    if (strcmp(s.md, noErrorCigar.c_str()) == 0) return;    // compares MD with "<L>M": never equal (quirk 5)
    std::vector<int> inserts(readLength, 0);
    int index = 0, curIndex = 0;
    walk_cigar(s.cigar, "IDMS^\t\n ", [&](char op, unsigned long n) {
        if (op == 'M') { index += (int)n; curIndex += (int)n; }
        else if (op == 'I' || op == 'S') {
            if (strandNo == 0) c.inPos[index]++; else c.inPos[readLength - index - 1]++;
            c.inLengths[n - 1]++;
            inserts[curIndex] = (int)n;
            index += (int)n;
        } else if (op == 'D') {
            if (strandNo == 0) c.delPos[index]++; else c.delPos[readLength - index - 1]++;
            c.delLengths[n - 1]++;
        }
    });
    walk_md(s.md, inserts, read, [&](int idx, int cur, char from, char to) {
        if (strandNo == 0) c.errorPos[idx - 1 + cur]++; else c.errorPos[readLength - idx - cur]++;
        int f = b5(from), t = b5(to);
        if (f != t) c.errorTypes[f][t]++;
    });
}

struct Probs {
    std::vector<double> errorPosDist, inPosDist, inLengthDist, delPosDist, delLengthDist, insertLengthDist, smoothed, noErrorProbs;
    double errorTypeProbs[5][5], baseErrorRates[5];
    double mean = 0, leftSD = 0, rightSD = 0;
};

void compute_probabilities(Counts &c, Probs &p) {        // computeProbabilites, :497-844
    for (int i = 0; i < 5; i++) {
        int errorCount = 0;
        for (int j = 0; j < 5; j++) errorCount += (int)c.errorTypes[i][j];
        for (int j = 0; j < 5; j++) p.errorTypeProbs[i][j] = (double)c.errorTypes[i][j] / errorCount;
        p.baseErrorRates[i] = errorCount / (double)c.baseCounts[i];
    }
    double sum = 0;
    for (int i = 0; i < 4; i++) sum += p.baseErrorRates[i];
    for (int i = 0; i < 4; i++) p.baseErrorRates[i] = 4 * p.baseErrorRates[i] / sum;
    p.baseErrorRates[4] = 1;
    int L = c.L;
    for (int i = L - 1; i > 0; i--) c.readLengths[i - 1] = c.readLengths[i] + c.readLengths[i - 1];
    p.errorPosDist.resize(L); p.inPosDist.resize(L); p.inLengthDist.resize(L); p.delPosDist.resize(L); p.delLengthDist.resize(L);
    for (int i = 0; i < L; i++) {
        p.errorPosDist[i] = (double)c.errorPos[i] / c.readLengths[i];
        p.inPosDist[i] = (double)c.inPos[i] / c.readLengths[i];
        p.delPosDist[i] = (double)c.delPos[i] / c.readLengths[i];
    }
    int inCount = 0, delCount = 0;
    for (int i = 0; i < L; i++) { inCount += (int)c.inLengths[i]; delCount += (int)c.delLengths[i]; }
    for (int i = 0; i < L; i++) { p.inLengthDist[i] = (double)c.inLengths[i] / inCount; p.delLengthDist[i] = (double)c.delLengths[i] / delCount; }
    int mis = c.maxInsertSize;
    p.insertLengthDist.resize(mis);
    long insCount = c.discardedReads;
    sum = 0;
    for (int i = 0; i < mis; i++) { insCount += (c.insertCounts[i] - 1); sum += i * (c.insertCounts[i] - 1); }
    p.mean = sum / insCount;
    for (int i = 0; i < mis; i++) p.insertLengthDist[i] = (double)c.insertCounts[i] / insCount;
    p.noErrorProbs.resize(L);
    double noErrorProb = 1.0;
    for (int i = 0; i < L; i++) { noErrorProb *= (1 - p.errorPosDist[i] - p.inPosDist[i] - p.delPosDist[i]); p.noErrorProbs[i] = noErrorProb; }
    // moving-average smoothing, window 12, edges copied, then the odd final correction (:646-677)
    const int W = 12;
    p.smoothed.resize(mis);
    double windowSum = 0;
    for (int i = 0; i < W; i++) p.smoothed[i] = p.insertLengthDist[i];
    for (int i = 0; i < 2 * W + 1; i++) windowSum += p.insertLengthDist[i];
    p.smoothed[W] = windowSum / (2 * W + 1);
    for (int i = W + 1; i < mis - W; i++) {
        windowSum -= p.insertLengthDist[i - W - 1];
        windowSum += p.insertLengthDist[i + W];
        p.smoothed[i] = windowSum / (2 * W + 1);
    }
    for (int i = mis - W; i < mis; i++) p.smoothed[i] = p.insertLengthDist[i];
    for (int i = 0; i < mis; i++) p.smoothed[i] = p.smoothed[i] - 1 / (double)(insCount) + (1 / (double)mis) / (double)(insCount + 1);
    // right / left SD about the mean (:785-802)
    double insertSum = 0, insertCount = 0;
    for (int i = (int)(p.mean + 1); i < mis; i++) {
        insertSum = insertSum + (c.insertCounts[i] - 1) * (i - p.mean) * (i - p.mean);
        insertCount += (c.insertCounts[i] - 1);
    }
    p.rightSD = sqrt(insertSum / insertCount);
    insertSum = 0; insertCount = 0;
    for (int i = std::max((int)(p.mean - 10 * p.rightSD), 0); i < p.mean; i++) {
        insertSum = insertSum + (c.insertCounts[i] - 1) * (p.mean - i) * (p.mean - i);
        insertCount += (c.insertCounts[i] - 1);
    }
    p.leftSD = sqrt(insertSum / insertCount);
}

long double error_prob(const Probs &p, const SamCols &s) {      // computeErrorProb, :952-1153 (long double)
    int strandNo = (s.flag & 16) >> 4;
    const char *read = s.seq;
    unsigned long readLength = strlen(read);
    long double errorProb = p.noErrorProbs[readLength - 1];
    if (s.md[5] == '^') return errorProb;
    std::vector<int> inserts(readLength, 0);
    int index = 0, curIndex = 0;
    walk_cigar(s.cigar, "IDM^\t\n ", [&](char op, unsigned long n) {
        if (op == 'M') { index += (int)n; curIndex += (int)n; }
        else if (op == 'I') {
            unsigned long i = strandNo == 0 ? (unsigned long)index : readLength - index - 1;
            errorProb = errorProb * p.inPosDist[i] * p.inLengthDist[n - 1] / (1 - p.errorPosDist[i] - p.inPosDist[i] - p.delPosDist[i]);
            inserts[curIndex] = (int)n;
            index += (int)n;
        } else if (op == 'D') {
            unsigned long i = strandNo == 0 ? (unsigned long)index : readLength - index - 1;
            errorProb = errorProb * p.delPosDist[i] * p.delLengthDist[n - 1] / (1 - p.errorPosDist[i] - p.inPosDist[i] - p.delPosDist[i]);
        }
    });
    walk_md(s.md, inserts, read, [&](int idx, int cur, char from, char to) {
        int i = strandNo == 0 ? idx - 1 + cur : (int)readLength - idx - cur;
        errorProb = errorProb * p.errorPosDist[i] / (1 - p.errorPosDist[i] - p.inPosDist[i] - p.delPosDist[i]);
        int f = b5(from), t = b5(to);
        if (f != t) errorProb *= p.baseErrorRates[f] * p.errorTypeProbs[f][t];
    });
    return errorProb;
}

}  // namespace

// ---- multi-threaded passes over myout.sam.  The reference re-parses the whole file twice in every one of
// its worker processes (Figbird.cpp:7110-7133); here the file is mapped once and both passes are split over host threads.
// Everything the passes accumulate is an integer count (position / type / length histograms, the gapProbs histogram), so the
// merged result is the sequential one exactly.  The one order-dependent corner -- an insert of exactly maxInsertSize grows the
// histogram and changes what later inserts do (:207-223) -- makes the caller fall back to the sequential pass.
namespace {

struct MapFile {
    const char *p = nullptr; size_t n = 0; int fd = -1;
    bool open(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; return false; }
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        p = (const char *)m;
        return true;
    }
    ~MapFile() { if (p && n) munmap((void *)p, n); if (fd >= 0) ::close(fd); }
};

inline size_t line_end(const MapFile &f, size_t b) { const void *e = memchr(f.p + b, '\n', f.n - b); return e ? (size_t)((const char *)e - f.p) + 1 : f.n; }
// copies the line at [b, e) as fgets(buf, kMaxRec) would return its first piece
inline void copy_line(const MapFile &f, size_t b, size_t e, char *buf) { size_t len = std::min(e - b, (size_t)kMaxRec - 1); memcpy(buf, f.p + b, len); buf[len] = 0; }
inline size_t next_line_start(const MapFile &f, size_t pos) { return pos == 0 ? 0 : line_end(f, pos - 1); }     // first line starting at or after pos

std::string qname_at(const MapFile &f, size_t b) { size_t e = b; while (e < f.n && f.p[e] != '\t' && f.p[e] != '\n') e++; return std::string(f.p + b, e - b); }

// start of the non-header line `back` lines before the line starting at b (b itself when back == 0); false if there are fewer
bool line_before(const MapFile &f, size_t b, int back, size_t &out) {
    size_t cur = b;
    for (int k = 0; k < back; ) {
        if (cur == 0) return false;
        size_t prev = cur - 1;                       // the newline that ends the previous line
        size_t s0 = prev;
        while (s0 > 0 && f.p[s0 - 1] != '\n') s0--;
        cur = s0;
        if (f.p[cur] != '@') k++;
    }
    out = cur;
    return true;
}

}  // namespace

bool build_model(const RunArgs &a, const Scaffold &sc, Model &out, std::string &err) {
    Counts c;
    long totalCount = 0, unCount = 0;
    {
        FILE *f = fopen((a.tmp + "stat.txt").c_str(), "r");
        if (!f) { err = "can't open stat.txt"; return false; }
        int ok = fscanf(f, "%ld %ld %d %d", &totalCount, &unCount, &c.L, &c.MAX_INSERT_SIZE);
        fclose(f);
        if (ok != 4 || c.L <= 0) { err = "malformed stat.txt"; return false; }
    }
    c.MAX_INSERT_SIZE = c.MAX_INSERT_SIZE > 20000 ? c.MAX_INSERT_SIZE : 20000;
    c.maxInsertSize = c.MAX_INSERT_SIZE;
    c.insertCounts.assign(c.maxInsertSize, 1);
    for (auto &r : c.errorTypes) for (long &v : r) v = 1;
    for (long &v : c.baseCounts) v = 1;
    c.errorPos.assign(c.L, 1); c.inPos.assign(c.L, 1); c.inLengths.assign(c.L, 1); c.delPos.assign(c.L, 1); c.delLengths.assign(c.L, 1);
    c.readLengths.assign(c.L, 0);
    std::string noErrorCigar = std::to_string(c.L) + "M";
    double inputMean = a.setinputmean == 1 ? a.isz : 0;

    MapFile mf;
    if (!mf.open(a.mapFile)) { err = "Can't open map file"; return false; }
    int T = 1;
    { const char *te = getenv("FIGFILL_THREADS"); unsigned hw = std::thread::hardware_concurrency(); T = te ? atoi(te) : (int)std::min<unsigned>(hw ? hw : 1, 16);
      if (T < 1) T = 1; if (mf.n < ((size_t)4 << 20) && !te) T = 1; }
    // line-aligned byte ranges
    std::vector<size_t> cut((size_t)T + 1, mf.n);
    for (int t = 0; t < T; t++) cut[t] = next_line_start(mf, mf.n / (size_t)T * (size_t)t);
    cut[0] = 0;

    // ---- pass 1 (Figbird.cpp:7110-7128): insert-size histogram + error position / type / length counts
    auto pass1 = [&](size_t b, size_t e, Counts &cc, bool allow_resize, bool &need_seq, std::string &perr) {
        char line[kMaxRec];
        for (size_t pos = b; pos < e; ) {
            size_t le = line_end(mf, pos);
            if (mf.p[pos] != '@') {
                copy_line(mf, pos, le, line);
                SamCols s; split_sam(line, s);
                if (s.ok && s.nh == 1 && s.md[0] && s.md[5] != '^') {
                    long contigNo = atol(s.rname);
                    if (contigNo < 0 || contigNo >= sc.n()) { perr = "myout.sam: contig index out of range"; return; }
                    if ((int)strlen(s.seq) > cc.L) { perr = "myout.sam: read longer than maxReadLength"; return; }
                    if ((double)(sc.off[contigNo + 1] - sc.off[contigNo]) > inputMean) {
                        if (!allow_resize && s.tlen >= cc.maxInsertSize && s.tlen <= cc.MAX_INSERT_SIZE) { need_seq = true; return; }
                        update_insert_counts(cc, s.tlen);
                    }
                    count_errors(cc, s, noErrorCigar);
                    cc.uniqueMappedReads++;
                }
            }
            pos = le;
        }
    };
    bool seq1 = T == 1;
    if (!seq1) {
        std::vector<Counts> tc((size_t)T);
        std::vector<char> need((size_t)T, 0); std::vector<std::string> errs((size_t)T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) {
            Counts &z = tc[(size_t)t];
            z.L = c.L; z.maxInsertSize = c.maxInsertSize; z.MAX_INSERT_SIZE = c.MAX_INSERT_SIZE;
            z.insertCounts.assign((size_t)c.maxInsertSize, 0);
            for (auto &r : z.errorTypes) for (long &v : r) v = 0;
            for (long &v : z.baseCounts) v = 0;
            z.errorPos.assign(c.L, 0); z.inPos.assign(c.L, 0); z.inLengths.assign(c.L, 0); z.delPos.assign(c.L, 0); z.delLengths.assign(c.L, 0); z.readLengths.assign(c.L, 0);
            th.emplace_back([&, t] { bool nd = false; pass1(cut[(size_t)t], cut[(size_t)t + 1], tc[(size_t)t], false, nd, errs[(size_t)t]); need[(size_t)t] = nd; });
        }
        for (auto &x : th) x.join();
        for (int t = 0; t < T; t++) if (!errs[(size_t)t].empty()) { err = errs[(size_t)t]; return false; }
        for (int t = 0; t < T; t++) if (need[(size_t)t]) seq1 = true;
        if (!seq1) {
            for (const Counts &z : tc) {
                for (size_t i = 0; i < z.insertCounts.size(); i++) c.insertCounts[i] += z.insertCounts[i];
                for (int i = 0; i < c.L; i++) { c.errorPos[i] += z.errorPos[i]; c.inPos[i] += z.inPos[i]; c.inLengths[i] += z.inLengths[i]; c.delPos[i] += z.delPos[i]; c.delLengths[i] += z.delLengths[i]; c.readLengths[i] += z.readLengths[i]; }
                for (int i = 0; i < 5; i++) { c.baseCounts[i] += z.baseCounts[i]; for (int j = 0; j < 5; j++) c.errorTypes[i][j] += z.errorTypes[i][j]; }
                c.discardedReads += z.discardedReads; c.uniqueMappedReads += z.uniqueMappedReads;
            }
        }
    }
    if (seq1) { bool nd = false; std::string perr; pass1(0, mf.n, c, true, nd, perr); if (!perr.empty()) { err = perr; return false; } }
    Probs p;
    compute_probabilities(c, p);

    // ---- pass 2: gapProbs histogram (computeLikelihood, :1156-1376); only what feeds gapProbCutOff is kept.  Ranges start on
    // a pair boundary where the (qname1, qname2) group changes; every group is counted when the next one starts -- the file's
    // LAST group never is (the reference only counts a group on seeing its successor, :1290-1330).
    std::vector<long> gapProbs(1000, 0);
    auto pass2 = [&](size_t b, size_t e, bool last_range, std::vector<long> &hist, bool &overflow) {
        char l1[kMaxRec], l2[kMaxRec];
        std::vector<long> effLen(c.maxInsertSize, -1);
        auto effective = [&](int ins) -> long {
            auto calc = [&](int v) { long ee = 0; for (int64_t i = 0; i < sc.n(); i++) { long cl = (long)(sc.off[i + 1] - sc.off[i]); if (cl >= v) ee += (cl - v + 1); } return ee; };
            if (ins < 0) return (long)sc.seq.size();
            if (ins >= c.maxInsertSize) return calc(ins);
            if (effLen[ins] == -1) effLen[ins] = calc(ins);
            return effLen[ins];
        };
        auto count_group = [&](long double sum, long double gapProb, long double &logsum) {
            if (sum < 1e-320 || std::isnan(sum)) sum = 1e-320;
            logsum += log10l(sum);
            int gapIndex = (int)(-log10l(gapProb));
            gapIndex++;
            if (gapIndex < 1000 && gapIndex >= 0) hist[gapIndex]++; else hist[999]++;
        };
        long double sum = 0, logsum = 0, gapProb = 0, tempProb = 0;
        std::string pre1 = "*", pre2 = "*";
        size_t pos = b;
        auto next = [&](char *buf) -> bool {           // next non-header line of the range
            while (pos < e) { size_t le = line_end(mf, pos); bool hdr = mf.p[pos] == '@'; if (!hdr) copy_line(mf, pos, le, buf); pos = le; if (!hdr) return true; }
            return false;
        };
        while (next(l1)) {
            if (!next(l2)) break;
            SamCols s1, s2; split_sam(l1, s1); split_sam(l2, s2);
            if (!s1.ok || !s2.ok) continue;
            int insertSize = std::max(s1.tlen, s2.tlen);
            long double insertSizeProb = 0;
            if (insertSize >= 0 && insertSize < c.maxInsertSize) insertSizeProb = p.insertLengthDist[insertSize];
            if (insertSizeProb == 0) insertSizeProb = 1 / (double)c.uniqueMappedReads;
            long double e1 = error_prob(p, s1), e2 = error_prob(p, s2);
            long double prob = (1 / (long double)(effective(insertSize))) * insertSizeProb * e1 * e2;
            bool same = pre1 == s1.qname && pre2 == s2.qname;
            if (same) {
                if (tempProb < prob) { tempProb = prob; gapProb = e2; }
                sum += prob;
            } else {
                if (pre1 != "*" && pre2 != "*") count_group(sum, gapProb, logsum);
                sum = prob; tempProb = prob; gapProb = e2;
            }
            pre1 = s1.qname; pre2 = s2.qname;
            if (std::isinf(logsum)) { overflow = true; return; }
        }
        if (!last_range && pre1 != "*" && pre2 != "*") count_group(sum, gapProb, logsum);      // its successor opens the next range
        if (std::isinf(logsum)) overflow = true;
    };
    {
        bool overflow = false;
        if (T == 1) pass2(0, mf.n, true, gapProbs, overflow);
        else {
            // pair alignment: global index of the first non-header line of each range, then forward to a group change
            std::vector<size_t> nlines((size_t)T, 0);
            {   std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back([&, t] { size_t k = 0; for (size_t pos = cut[(size_t)t]; pos < cut[(size_t)t + 1]; pos = line_end(mf, pos)) if (mf.p[pos] != '@') k++; nlines[(size_t)t] = k; });
                for (auto &x : th) x.join(); }
            std::vector<size_t> pc((size_t)T + 1, mf.n);
            size_t idx = 0;
            pc[0] = 0;
            for (int t = 1; t < T; t++) {
                idx += nlines[(size_t)t - 1];
                size_t P = cut[(size_t)t];
                auto skip_hdr = [&](size_t q) { while (q < mf.n && mf.p[q] == '@') q = line_end(mf, q); return q; };
                P = skip_hdr(P);
                if (idx & 1) { if (P < mf.n) P = skip_hdr(line_end(mf, P)); }
                while (P < mf.n) {                       // forward to the first pair that opens a new (qname1, qname2) group
                    size_t pp;
                    if (!line_before(mf, P, 2, pp)) break;
                    size_t pp2 = skip_hdr(line_end(mf, pp));
                    size_t P2 = skip_hdr(line_end(mf, P));
                    if (P2 >= mf.n) break;
                    if (qname_at(mf, pp) == qname_at(mf, P) && qname_at(mf, pp2) == qname_at(mf, P2)) P = skip_hdr(line_end(mf, P2));
                    else break;
                }
                pc[(size_t)t] = std::max(P, pc[(size_t)t - 1]);
            }
            std::vector<std::vector<long>> hs((size_t)T, std::vector<long>(1000, 0));
            std::vector<char> ov((size_t)T, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] { bool o = false; bool last = true; for (int u = t + 1; u < T; u++) if (pc[(size_t)u] < mf.n) last = false;
                                                               pass2(pc[(size_t)t], pc[(size_t)t + 1], last, hs[(size_t)t], o); ov[(size_t)t] = o; });
            for (auto &x : th) x.join();
            for (int t = 0; t < T; t++) { if (ov[(size_t)t]) overflow = true; for (int i = 0; i < 1000; i++) gapProbs[i] += hs[(size_t)t][i]; }
        }
        if (overflow) { err = "model likelihood overflow (reference exits here)"; return false; }
    }
    long gapProbSum = 0, gapProbCount = 0;
    for (long v : gapProbs) gapProbSum += v;
    int cutoff = 0;
    for (int i = 0; i < 1000; i++) { gapProbCount += gapProbs[i]; if (gapProbCount >= .8 * gapProbSum) { cutoff = i; break; } }

    out.errorPosDist = p.errorPosDist; out.inPosDist = p.inPosDist; out.delPosDist = p.delPosDist;
    out.insertLengthDistSmoothed = p.smoothed;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) out.errorTypeProbs[i * 5 + j] = p.errorTypeProbs[i][j];
    out.maxReadLength = c.L; out.maxInsertSize = c.maxInsertSize; out.cutoff = cutoff;
    out.insertSizeMean = p.mean; out.leftSD = p.leftSD; out.rightSD = p.rightSD;
    out.Tmin = std::max((int)(p.mean - 3 * p.leftSD), 1);                       // :7193-7200
    out.Tmax = std::min((int)(p.mean + 3 * p.rightSD), c.maxInsertSize);
    if (a.partial_flag) { out.Tmin -= a.partial_len; out.Tmax += a.partial_len; }
    return true;
}

void Model::fill(fig_model &m, const RunArgs &a) const {
    memset(&m, 0, sizeof(m));
    m.max_read_length = maxReadLength;
    m.error_pos_dist = errorPosDist.data(); m.in_pos_dist = inPosDist.data(); m.del_pos_dist = delPosDist.data();
    for (int i = 0; i < 25; i++) m.error_type_probs[i] = errorTypeProbs[i];
    m.insert_len_dist_smoothed = insertLengthDistSmoothed.data(); m.max_insert_size = maxInsertSize;
    m.insert_threshold_min = Tmin; m.insert_threshold_max = Tmax; m.gap_prob_cutoff = cutoff;
    m.partial_flag = a.partial_flag; m.unmapped_flag = a.unmapped; m.script_itr = a.script_itr; m.max_distance = a.D;
    m.read_length = a.read_length; m.neg_overlap = a.neg_overlap; m.partial_len = a.partial_len; m.unm_limit = a.unm_limit;
}

// ------------------------------------------------------------------ per-gap inputs
static bool read_lines(const std::string &path, std::vector<std::string> &out) {
    out.clear();
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[kMaxRec];
    while (fgets(buf, kMaxRec, f)) out.emplace_back(buf);
    fclose(f);
    return true;
}

static char comp(char ch) {                 // reverse(), Figbird.cpp:1427-1449
    switch (ch) { case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A'; default: return 'N'; }
}

// per-gap line source: (is_partial, gap, lines) -> found?
typedef std::function<bool(bool, size_t, std::vector<std::string> &)> GapLines;
static bool load_batch_src(const RunArgs &a, const Scaffold &sc, const GapLines &src, Batch &B, std::string &err);

bool load_batch(const RunArgs &a, const Scaffold &sc, Batch &B, std::string &err) {
    return load_batch_src(a, sc, [&](bool partial, size_t g, std::vector<std::string> &lines) {
        return read_lines(a.gapsDir + (partial ? "partial_gaps_" : "gaps_") + std::to_string(g) + ".sam", lines); }, B, err);
}

static void split_lines(const std::string &text, std::vector<std::string> &out) {      // as fgets() would cut them (kMaxRec-1 characters at most)
    out.clear();
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        size_t end = e == std::string::npos ? text.size() : e + 1;
        while (end - p > (size_t)kMaxRec - 1) { out.emplace_back(text, p, (size_t)kMaxRec - 1); p += (size_t)kMaxRec - 1; }
        out.emplace_back(text, p, end - p);
        p = end;
    }
}

bool load_batch_mem(const RunArgs &a, const Scaffold &sc, const std::vector<std::string> *gaps_text, const std::vector<std::string> *partial_text,
                    Batch &B, std::string &err) {
    return load_batch_src(a, sc, [&](bool partial, size_t g, std::vector<std::string> &lines) {
        const std::vector<std::string> *t = partial ? partial_text : gaps_text;
        lines.clear();
        if (!t) return partial;                         // no partial texts: empty files (the reference needs them to exist); no gaps texts: missing
        if (g < t->size()) split_lines((*t)[g], lines);
        return true; }, B, err);
}

static bool load_batch_src(const RunArgs &a, const Scaffold &sc, const GapLines &src, Batch &B, std::string &err) {
    std::vector<std::string> gi, st2;
    if (!read_lines(a.tmp + "gapInfo.txt", gi)) { err = "Couldn't open gapinfo"; return false; }
    if (!read_lines(a.tmp + "stat2.txt", st2)) { err = "Couldn't open stat2.txt"; return false; }
    size_t ng = std::min(gi.size(), st2.size());        // the per-gap loop reads both files in lock step (:7329)
    B = Batch();
    B.u_read_off.push_back(0); B.p_read_off.push_back(0); B.u_seq_off.push_back(0); B.p_seq_off.push_back(0);
    for (size_t g = 0; g < ng; g++) {
        std::vector<char> b1(gi[g].begin(), gi[g].end()); b1.push_back(0);
        char *sv = nullptr;
        char *t = strtok_r(b1.data(), "\t", &sv); int contig = t ? atoi(t) : 0;
        t = strtok_r(nullptr, "\t", &sv); long start = t ? atol(t) : 0;
        t = strtok_r(nullptr, "\t\n", &sv); int len = t ? atoi(t) : 0;
        if (contig < 0 || contig >= sc.n()) { err = "gapInfo.txt: contig index out of range"; return false; }
        B.gap_contig.push_back(contig); B.gap_start.push_back(start); B.gap_len.push_back(len);
        std::vector<char> b2(st2[g].begin(), st2[g].end()); b2.push_back(0);
        sv = nullptr;
        for (int k = 0; k < 3; k++) { t = strtok_r(k == 0 ? b2.data() : nullptr, "\t", &sv); B.gap_stat2.push_back(t ? atoi(t) : 0); }
        int fillflag = 1;
        std::string gs = std::to_string(g);
        std::vector<std::string> lines;
        if (a.unmapped == 1) {
            // findcount_file(...,0) :6686-6711 then parseUnmapped :5661-5767
            if (!src(false, g, lines)) { err = "missing gaps_" + gs + ".sam"; return false; }
            long pairs = (long)(lines.size() / 2);
            int r_count1 = (int)pairs;
            if (pairs > kReadCap) { B.messages.push_back("Gap = " + gs + "\tReads = " + std::to_string(pairs)); r_count1 = kReadCap; fillflag = -1; }
            int r_c = 0, total_read = 0;
            size_t li = 0;
            while (li < lines.size()) {
                const std::string &l1 = lines[li++];
                if (!l1.empty() && l1[0] == '@') continue;
                if (l1.size() < 60) { B.messages.push_back("unknown problem in sam, skipping read no - " + std::to_string(r_c) + " for gap - " + gs); continue; }
                if (li >= lines.size()) break;
                const std::string &l2 = lines[li++];
                if (r_c >= r_count1) break;
                std::vector<char> c1(l1.begin(), l1.end()); c1.push_back(0);
                std::vector<char> c2(l2.begin(), l2.end()); c2.push_back(0);
                sv = nullptr;
                strtok_r(c1.data(), "\t", &sv);
                t = strtok_r(nullptr, "\t", &sv); int flag = t ? atoi(t) : 0;
                strtok_r(nullptr, "\t", &sv);
                t = strtok_r(nullptr, "\t", &sv); int pos = t ? atoi(t) : 0;
                sv = nullptr;
                strtok_r(c2.data(), "\t", &sv);
                for (int k = 0; k < 5; k++) strtok_r(nullptr, "\t", &sv);
                char *rs = strtok_r(nullptr, "\t", &sv);
                std::string seq = rs ? rs : "";
                while (!seq.empty() && (seq.back() == '\n' || seq.back() == '\r')) seq.pop_back();
                bool fwd_anchor = ((flag & 16) >> 4) == 0;
                if (fwd_anchor) { std::string rc(seq.size(), 'N'); for (size_t i = 0; i < seq.size(); i++) rc[seq.size() - 1 - i] = comp(seq[i]); seq = rc; }
                if (seq.empty()) { err = "gaps_" + gs + ".sam: empty mate sequence"; return false; }
                B.u_anchor_pos.push_back(pos); B.u_is_reverse.push_back(fwd_anchor ? 1 : 0);
                B.u_seq += seq; B.u_seq_off.push_back((int64_t)B.u_seq.size());
                r_c++; total_read++;
                if (total_read == kReadCap) break;
            }
        }
        B.u_read_off.push_back((int64_t)B.u_anchor_pos.size());
        if (!src(true, g, lines)) { err = "missing partial_gaps_" + gs + ".sam (the reference fopen()s it unconditionally)"; return false; }
        size_t keep = std::min(lines.size(), (size_t)kReadCap + 1);
        for (size_t k = 0; k < keep; k++) {
            std::vector<char> c1(lines[k].begin(), lines[k].end()); c1.push_back(0);
            sv = nullptr;
            char *seq = strtok_r(c1.data(), "\t", &sv);
            t = strtok_r(nullptr, "\t", &sv); int clip = t ? atoi(t) : 0;
            t = strtok_r(nullptr, "\t", &sv); int match = t ? atoi(t) : 0;
            t = strtok_r(nullptr, "\t", &sv); int pos = t ? atoi(t) : 0;
            strtok_r(nullptr, "\t", &sv);
            t = strtok_r(nullptr, "\t", &sv); int ref = t ? atoi(t) : 0;
            char *q = strtok_r(nullptr, "\t", &sv);
            std::string s = seq ? seq : "", qs = q ? q : "";
            while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
            while (!qs.empty() && (qs.back() == '\n' || qs.back() == '\r')) qs.pop_back();
            if (s.empty()) { err = "partial_gaps_" + gs + ".sam: empty sequence"; return false; }
            qs.resize(s.size(), 'I');
            B.p_clipped_index.push_back(clip); B.p_match.push_back(match); B.p_pos.push_back(pos); B.p_ref_pos.push_back(ref);
            B.p_seq += s; B.p_qual += qs; B.p_seq_off.push_back((int64_t)B.p_seq.size());
        }
        B.p_read_off.push_back((int64_t)B.p_clipped_index.size());
        B.gap_fillflag.push_back(fillflag);
    }
    assign_processes(a, sc, B);
    return true;
}

// ------------------------------------------------------------------ the reference's worker processes
std::vector<std::vector<int32_t>> thread_allocation(const std::vector<int32_t> &gap_len, int num_threads, int gapthresh) {
    const int tot = (int)gap_len.size();
    std::vector<std::vector<int32_t>> T;
    if (tot == 0) return T;
    if (num_threads < 1) num_threads = 1;
    if (tot <= num_threads) {                                     // :459-463, :493-500: one gap per process
        T.resize((size_t)tot);
        for (int i = 0; i < tot; i++) T[(size_t)i].push_back(i);
        return T;
    }
    int per;                                                      // :466-471 (float, then rounded up)
    { const float tf = (float)(tot * 1.0 / num_threads); per = (int)tf; if (tf - (float)per > 0) per++; }
    T.resize((size_t)num_threads);
    std::vector<int32_t> small, large;                            // :505-531
    for (int g = 0; g < tot; g++) (gap_len[(size_t)g] > gapthresh ? large : small).push_back(g);
    // short gaps: round-robin over the processes (:538-579)
    for (size_t k = 0; k < small.size(); k++) T[k % (size_t)num_threads].push_back(small[k]);
    // long gaps: processes in descending order of remaining capacity (std::sort on {process, remaining}, as :586-603)
    std::vector<std::vector<int>> rem((size_t)num_threads, std::vector<int>(2, 0));
    for (int i = 0; i < num_threads; i++) { rem[(size_t)i][0] = i; rem[(size_t)i][1] = per - (int)T[(size_t)i].size(); }
    std::sort(rem.begin(), rem.end(), [](const std::vector<int> &x, const std::vector<int> &y) { return x[1] > y[1]; });
    size_t done = 0;
    if ((int)large.size() <= num_threads) {                       // :611-626: one each, in that order
        for (int i = 0; i < num_threads && done < large.size(); i++) T[(size_t)rem[(size_t)i][0]].push_back(large[done++]);
    } else {                                                      // :628-647: fill each process up to its capacity
        for (int i = 0; i < num_threads && done < large.size(); i++)
            for (int j = 0; j < rem[(size_t)i][1] && done < large.size(); j++) T[(size_t)rem[(size_t)i][0]].push_back(large[done++]);
    }
    for (auto &t : T) std::sort(t.begin(), t.end());              // writeGapLoad :322-323
    return T;
}

void assign_processes(const RunArgs &a, const Scaffold &, Batch &B) {
    B.processes = thread_allocation(B.gap_len, a.num_threads, a.unm_limit);
    B.gap_ot_preset.assign(B.gap_contig.size(), 0);               // until measured: ot_presets_from_reach
}

// The carry of the reference's process-global overlap_threshold (Figbird.cpp:103, :6317) from the measured bits
// (fig_batch_probe_reach: reach[g] = the candidate loop of gap g gets to :6317): along each worker process's gap list.
void ot_presets_from_reach(Batch &B, const uint8_t *reach) {
    B.gap_ot_preset.assign(B.gap_contig.size(), 0);
    for (const auto &p : B.processes) fig_ot_carry(p, reach, B.gap_ot_preset.data());
}

bool write_gaploads(const RunArgs &a, const Batch &b, std::string &err) {
    FILE *f = fopen((a.tmp + "gaploads.txt").c_str(), "w");
    if (!f) { err = "can't write gaploads.txt"; return false; }
    for (const auto &p : b.processes) { for (int32_t g : p) fprintf(f, "%d\t", g); fprintf(f, "\n"); }
    fclose(f);
    return true;
}

void Batch::view(fig_gap_batch &b, const Scaffold &sc) const {
    memset(&b, 0, sizeof(b));
    b.n_gaps = (int64_t)gap_contig.size();
    b.n_contigs = sc.n(); b.contig_off = sc.off.data(); b.contig_seq = sc.seq.data();
    b.gap_contig = gap_contig.data(); b.gap_start = gap_start.data(); b.gap_len = gap_len.data();
    b.gap_stat2 = gap_stat2.data(); b.gap_fillflag = gap_fillflag.data();
    b.u_read_off = u_read_off.data(); b.u_anchor_pos = u_anchor_pos.data(); b.u_is_reverse = u_is_reverse.data();
    b.u_seq_off = u_seq_off.data(); b.u_seq = u_seq.data();
    b.p_read_off = p_read_off.data(); b.p_clipped_index = p_clipped_index.data(); b.p_match = p_match.data();
    b.p_pos = p_pos.data(); b.p_ref_pos = p_ref_pos.data(); b.p_seq_off = p_seq_off.data(); b.p_seq = p_seq.data(); b.p_qual = p_qual.data();
    b.gap_ot_preset = gap_ot_preset.size() == gap_contig.size() && !gap_ot_preset.empty() ? gap_ot_preset.data() : nullptr;
}

// ------------------------------------------------------------------ outputs
bool write_gapout(const RunArgs &a, const Batch &b, const Results &r, std::string &err) {
    FILE *f = fopen((a.tmp + "gapout.txt").c_str(), "w");
    if (!f) { err = "can't write gapout.txt"; return false; }
    for (size_t g = 0; g < b.gap_contig.size(); g++) {
        int n = r.filled_len[g];
        fprintf(f, "%d\t%d\t%ld\t%d\t%d\t", (int)g, b.gap_contig[g], (long)b.gap_start[g], b.gap_len[g], n);
        if (n > 0) fwrite(r.str.data() + r.str_off[g], 1, (size_t)n, f);
        fputc('\n', f);
    }
    fclose(f);
    return true;
}

bool write_draw(const RunArgs &a, const Batch &b, const Results &r, std::string &err) {
    FILE *f = fopen((a.tmp + "draw.txt").c_str(), "w");
    if (!f) { err = "can't write draw.txt"; return false; }
    if (r.draw_len.empty()) { fclose(f); return true; }
    int readlen = a.read_length;
    int64_t nu = (int64_t)b.u_anchor_pos.size();
    auto header = [&](int g, int length) {
        for (int i = 0; i < readlen; i++) fputc(' ', f);
        fprintf(f, "====================+Gap = %d starting,length = %d===============================\n", g, length);
        for (int i = 0; i < readlen; i++) fputc(' ', f);
        for (int i = 0; i < length; i++) fputc('N', f);
        fputc('\n', f);
    };
    auto readline = [&](const char *s, int slen, int readno, int length, int isz, char type) {
        for (int i = 0; i < readlen + length; i++) fputc(' ', f);
        fwrite(s, 1, (size_t)slen, f);
        fprintf(f, "[%d %d isz = %d %c]\n", readno, length, isz, type);
    };
    // the reference concatenates the worker processes' draw files in process order (mergeFiles, FillGaps.cpp:222-260)
    std::vector<size_t> order;
    for (const auto &p : b.processes) for (int32_t g : p) order.push_back((size_t)g);
    if (order.size() != b.gap_contig.size()) { order.clear(); for (size_t g = 0; g < b.gap_contig.size(); g++) order.push_back(g); }
    for (size_t g : order) {
        int lu = r.draw_len[g * 2], lp = r.draw_len[g * 2 + 1];
        if (lu >= 0) {
            header((int)g, lu);
            int gapoffset = lu - b.gap_len[g];
            for (int64_t k = b.u_read_off[g]; k < b.u_read_off[g + 1]; k++) {
                if (r.draw_pos[k] == INT32_MIN) continue;
                long pos1 = b.u_anchor_pos[k];
                if (!(pos1 < b.gap_start[g])) pos1 += gapoffset;
                char type = pos1 < b.gap_start[g] ? 'I' : 'E';
                readline(b.u_seq.data() + b.u_seq_off[k], (int)(b.u_seq_off[k + 1] - b.u_seq_off[k]), (int)(k - b.u_read_off[g]), r.draw_pos[k], r.draw_isz[k], type);
            }
        }
        if (lp >= 0) {
            header((int)g, lp);
            for (int64_t k = b.p_read_off[g]; k < b.p_read_off[g + 1]; k++) {
                if (r.draw_pos[nu + k] == INT32_MIN) continue;
                readline(b.p_seq.data() + b.p_seq_off[k], (int)(b.p_seq_off[k + 1] - b.p_seq_off[k]), (int)(k - b.p_read_off[g]), r.draw_pos[nu + k], r.draw_isz[nu + k], 'P');
            }
        }
    }
    fclose(f);
    return true;
}

bool write_scaffold(const RunArgs &a, const Scaffold &sc, const Batch &b, const Results &r, std::string &err) {
    FILE *out = fopen((a.tmp + "filledContigs.fa").c_str(), "w");
    FILE *nf = fopen((a.tmp + "Ncount.txt").c_str(), "w");
    if (!out || !nf) { err = "can't write filledContigs.fa / Ncount.txt"; if (out) fclose(out); if (nf) fclose(nf); return false; }
    std::vector<int> gtf(r.gaptofill.begin(), r.gaptofill.end());      // consumed like FillGaps.cpp's array (:900-904)
    size_t ng = b.gap_contig.size();
    int nStart = 0, gap_count = -1;
    long newNcount = 0;
    size_t next_gap = 0;
    std::string gapString;              // keeps the previous gap's string when a gap closes to length 0 (:861-870)
    int gapStringLength = 0;
    std::string buf;
    for (int64_t i = 0; i < sc.n(); i++) {
        const char *cs = sc.seq.data() + sc.off[i];
        int64_t n = sc.off[i + 1] - sc.off[i];
        fprintf(out, ">%s\n", sc.names[i].c_str());
        buf.clear();
        for (int64_t j = 0; j < n; j++) {
            bool isN = cs[j] == 'N' || cs[j] == 'n';
            if (isN && nStart == 0) { nStart = 1; gap_count++; }
            if (!isN || j == n - 1) {
                if (nStart == 1) {
                    if (next_gap < ng) {
                        gapStringLength = r.filled_len[next_gap];
                        if (gapStringLength > 0) gapString.assign(r.str.data() + r.str_off[next_gap], (size_t)gapStringLength);
                        next_gap++;
                    }
                    for (char ch : gapString) if (ch == 'N') newNcount++;
                    fwrite(buf.data(), 1, buf.size(), out);
                    if (gapStringLength > 0) fwrite(gapString.data(), 1, gapString.size(), out);
                    buf.clear();
                    nStart = 0;
                }
                if (gap_count >= 0 && gap_count < (int)gtf.size() && gtf[gap_count] > 0) gtf[gap_count]--;
                else buf.push_back(cs[j]);
            }
        }
        fwrite(buf.data(), 1, buf.size(), out);
        fputc('\n', out);
    }
    fprintf(nf, "%d", newNcount == 0 ? 0 : 1);                          // FillGaps.cpp:920-922
    fclose(out); fclose(nf);
    return true;
}

}  // namespace fighost

// ------------------------------------------------------------------ C wrappers (libfighost.so)
// Used by the Python tests / bench to obtain the run-level model exactly as figfill builds it.
namespace fighost {

void make_shard(const Batch &B, const std::vector<int64_t> &ids, Batch &S) {
    S = Batch();
    S.u_read_off.push_back(0); S.p_read_off.push_back(0); S.u_seq_off.push_back(0); S.p_seq_off.push_back(0);
    const size_t ng = B.gap_contig.size();
    for (int64_t g : ids) {
        S.gap_contig.push_back(B.gap_contig[g]); S.gap_start.push_back(B.gap_start[g]); S.gap_len.push_back(B.gap_len[g]);
        for (int q = 0; q < 3; q++) S.gap_stat2.push_back(B.gap_stat2[g * 3 + q]);
        S.gap_fillflag.push_back(B.gap_fillflag[g]);
        S.gap_ot_preset.push_back(B.gap_ot_preset.size() == ng ? B.gap_ot_preset[g] : 0);
        for (int64_t i = B.u_read_off[g]; i < B.u_read_off[g + 1]; i++) {
            S.u_anchor_pos.push_back(B.u_anchor_pos[i]); S.u_is_reverse.push_back(B.u_is_reverse[i]);
            S.u_seq.append(B.u_seq, (size_t)B.u_seq_off[i], (size_t)(B.u_seq_off[i + 1] - B.u_seq_off[i]));
            S.u_seq_off.push_back((int64_t)S.u_seq.size());
        }
        S.u_read_off.push_back((int64_t)S.u_anchor_pos.size());
        for (int64_t i = B.p_read_off[g]; i < B.p_read_off[g + 1]; i++) {
            S.p_clipped_index.push_back(B.p_clipped_index[i]); S.p_match.push_back(B.p_match[i]); S.p_pos.push_back(B.p_pos[i]); S.p_ref_pos.push_back(B.p_ref_pos[i]);
            S.p_seq.append(B.p_seq, (size_t)B.p_seq_off[i], (size_t)(B.p_seq_off[i + 1] - B.p_seq_off[i]));
            S.p_qual.append(B.p_qual, (size_t)B.p_seq_off[i], (size_t)(B.p_seq_off[i + 1] - B.p_seq_off[i]));
            S.p_seq_off.push_back((int64_t)S.p_seq.size());
        }
        S.p_read_off.push_back((int64_t)S.p_clipped_index.size());
    }
}

std::vector<double> estimate_cost(const Batch &b, const RunArgs &a, int L) {
    const size_t ng = b.gap_len.size();
    std::vector<double> c(ng, 1.0);
    for (size_t g = 0; g < ng; g++) {
        const double G = b.gap_len[g];
        if (a.unmapped == 1) {
            const double R = (double)(b.u_read_off[g + 1] - b.u_read_off[g]);
            const double cand = G <= a.unm_limit / 3 ? 3.0 * a.partial_len - 0.3 * G : (G <= a.unm_limit ? 2.0 * G : 1.0);
            const double its = G <= a.unm_limit / 3 ? 10.5 : (G <= a.unm_limit ? 11.5 : 2.0);      // placeReads calls per candidate (tools/cost_model_check.py)
            const double Gc = G <= a.unm_limit / 3 ? 0.5 * (0.3 * G + 3.0 * a.partial_len) : (G <= a.unm_limit ? 1.5 * G : G);   // typical candidate length
            const double W = std::min(Gc + L, 2200.0);
            c[g] = R * W * L * std::max(cand, 1.0) * its + 1.0;
        } else {
            const double R = (double)(b.p_read_off[g + 1] - b.p_read_off[g]);
            const double cand = G <= a.partial_len ? 3.0 * a.partial_len : (G <= 2 * a.partial_len ? 5.0 * G : 1.0);
            c[g] = R * (double)L * L * std::max(cand, 1.0) * 3.0 + 1.0;
        }
    }
    return c;
}

std::vector<std::vector<int64_t>> partition_lpt(const std::vector<double> &cost, int world) {
    std::vector<int64_t> order(cost.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int64_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return cost[x] > cost[y]; });
    std::vector<double> load((size_t)world, 0.0);
    std::vector<std::vector<int64_t>> bins((size_t)world);
    for (int64_t g : order) {
        size_t r = 0;
        for (size_t k = 1; k < load.size(); k++) if (load[k] < load[r]) r = k;
        bins[r].push_back(g); load[r] += cost[g];
    }
    for (auto &b : bins) std::sort(b.begin(), b.end());
    return bins;
}

}  // namespace fighost

extern "C" int fighost_build_model(const char *contig_file, const char *tmp_dir, const char *map_file, int partial_flag,
                                   int partial_len, int setinputmean, int isz, int max_read_cap, int insd_cap,
                                   double *e, double *ins, double *del, double *T25, double *insd, int *ints5, double *stats3) {
    fighost::RunArgs a;
    a.contigFile = contig_file; a.tmp = tmp_dir; a.mapFile = map_file; a.partial_flag = partial_flag; a.partial_len = partial_len;
    a.setinputmean = setinputmean; a.isz = isz;
    std::string err;
    fighost::Scaffold sc;
    if (!fighost::load_scaffold(a.contigFile, sc, err)) { fprintf(stderr, "%s\n", err.c_str()); return -1; }
    fighost::Model M;
    if (!fighost::build_model(a, sc, M, err)) { fprintf(stderr, "%s\n", err.c_str()); return -1; }
    if (M.maxReadLength > max_read_cap || M.maxInsertSize > insd_cap) return -2;
    for (int k = 0; k < M.maxReadLength; k++) { e[k] = M.errorPosDist[k]; ins[k] = M.inPosDist[k]; del[k] = M.delPosDist[k]; }
    for (int i = 0; i < 25; i++) T25[i] = M.errorTypeProbs[i];
    for (int i = 0; i < M.maxInsertSize; i++) insd[i] = M.insertLengthDistSmoothed[i];
    ints5[0] = M.maxReadLength; ints5[1] = M.maxInsertSize; ints5[2] = M.Tmin; ints5[3] = M.Tmax; ints5[4] = M.cutoff;
    stats3[0] = M.insertSizeMean; stats3[1] = M.leftSD; stats3[2] = M.rightSD;
    return 0;
}

// ------------------------------------------------------------------ run handle (libfighost.so)
// One gap-fill run as an object, for the multi-GPU launcher (figbird_amd/figfill_mp.py): every rank opens the run (same
// inputs, same model), fills a SHARD of the gap set through the C ABI, and rank 0 writes the four output files from the
// all-gathered results.  This is the role of FillGaps.cpp:456-649 (gap -> worker allocation), :668-679 (fan-out) and
// :688-926 (merge + scaffold rebuild) with GPUs in place of `system("g++ Figbird.cpp ...")` workers.
namespace {
struct Run {
    fighost::RunArgs a;
    fighost::Scaffold sc;
    fighost::Batch B, sub;
    fighost::Model M;
};
void set_err(char *err, int cap, const std::string &m) { if (err && cap > 0) { snprintf(err, (size_t)cap, "%s", m.c_str()); } }
}  // namespace

extern "C" void *fighost_run_open(const char *const *argv15, char *err, int errcap) {
    Run *r = new Run();
    fighost::RunArgs &a = r->a;
    a.contigFile = argv15[0]; a.D = atoi(argv15[1]); a.read_length = atoi(argv15[2]); a.script_itr = atoi(argv15[3]);
    a.partial_flag = atoi(argv15[4]); a.unmapped = atoi(argv15[5]); a.num_threads = atoi(argv15[6]); a.mapFile = argv15[7];
    a.tmp = argv15[8]; a.gapsDir = argv15[9]; a.neg_overlap = atoi(argv15[10]); a.partial_len = atoi(argv15[11]);
    a.trim = atoi(argv15[12]); a.setinputmean = atoi(argv15[13]); a.isz = atoi(argv15[14]);
    a.unm_limit = 400;
    std::string e;
    if (!fighost::load_scaffold(a.contigFile, r->sc, e) || !fighost::load_batch(a, r->sc, r->B, e) || !fighost::build_model(a, r->sc, r->M, e)) {
        set_err(err, errcap, e); delete r; return nullptr;
    }
    return r;
}
extern "C" void fighost_run_close(void *h) { delete (Run *)h; }
extern "C" int64_t fighost_run_ngaps(void *h) { return (int64_t)((Run *)h)->B.gap_contig.size(); }
extern "C" int fighost_run_sizes(void *h, int32_t *gap_len, int64_t *n_u, int64_t *n_p) {
    const fighost::Batch &B = ((Run *)h)->B;
    for (size_t g = 0; g < B.gap_contig.size(); g++) {
        gap_len[g] = B.gap_len[g]; n_u[g] = B.u_read_off[g + 1] - B.u_read_off[g]; n_p[g] = B.p_read_off[g + 1] - B.p_read_off[g];
    }
    return 0;
}
extern "C" int fighost_run_params(void *h, int32_t *out6) {        // read length, partial_len, unmapped flag, unm_limit, messages, -
    const Run *r = (Run *)h;
    out6[0] = r->M.maxReadLength; out6[1] = r->a.partial_len; out6[2] = r->a.unmapped; out6[3] = r->a.unm_limit; out6[4] = (int32_t)r->B.messages.size(); out6[5] = 0;
    return 0;
}
extern "C" const char *fighost_run_message(void *h, int i) { return ((Run *)h)->B.messages[(size_t)i].c_str(); }
extern "C" int fighost_run_model(void *h, fig_model *out) { Run *r = (Run *)h; r->M.fill(*out, r->a); return 0; }

// reach[g] for every gap of the run (global gap ids; figfill_mp all-gathers the shards' fig_batch_probe_reach bits) ->
// preset[g] = the gap's worker process of the reference has set overlap_threshold before it gets to the gap.
extern "C" int fighost_run_ot_presets(void *h, const uint8_t *reach, uint8_t *preset) {
    Run *r = (Run *)h;
    fighost::ot_presets_from_reach(r->B, reach);
    for (size_t g = 0; g < r->B.gap_ot_preset.size(); g++) preset[g] = r->B.gap_ot_preset[g];
    return 0;
}

// Sub-batch of the gaps `ids` (any order; kept in that order), owned by the handle until the next call.
extern "C" int fighost_run_shard(void *h, const int64_t *ids, int64_t n, fig_gap_batch *out, int64_t *n_ureads, int64_t *n_preads) {
    Run *r = (Run *)h;
    const int64_t ng = (int64_t)r->B.gap_contig.size();
    std::vector<int64_t> v(ids, ids + n);
    for (int64_t g : v) if (g < 0 || g >= ng) return -1;
    fighost::make_shard(r->B, v, r->sub);
    r->sub.view(*out, r->sc);
    *n_ureads = (int64_t)r->sub.u_anchor_pos.size(); *n_preads = (int64_t)r->sub.p_clipped_index.size();
    return 0;
}

// Which of `world` shards the C++ host (figfill with FIGFILL_DEVICES) puts every gap of the run `argv15` on: owner[g] = rank.
// Returns the number of gaps, or -1.  (Lets the tests pin the C++ deal to the Python one of figfill_mp.)
extern "C" int64_t fighost_partition(const char *const *argv15, int world, int32_t *owner) {
    char err[256];
    Run *r = (Run *)fighost_run_open(argv15, err, sizeof(err));
    if (!r) return -1;
    const int64_t ng = (int64_t)r->B.gap_contig.size();
    std::vector<std::vector<int64_t>> sh = fighost::partition_lpt(fighost::estimate_cost(r->B, r->a, r->M.maxReadLength), world);
    for (int k = 0; k < world; k++) for (int64_t g : sh[k]) owner[g] = k;
    fighost_run_close(r);
    return ng;
}

// gapout.txt, draw.txt, filledContigs.fa, Ncount.txt of the WHOLE gap set from arrays in global gap / read order
// (draw_*: [all unmapped reads..., all partial reads...] as in fig_gap_results).
extern "C" int fighost_run_write(void *h, const int32_t *filled_len, const int32_t *gaptofill, const int64_t *str_off, const char *str,
                                 const int32_t *draw_pos, const int32_t *draw_isz, const int32_t *draw_len, char *err, int errcap) {
    Run *r = (Run *)h;
    const size_t ng = r->B.gap_contig.size();
    const size_t nr = r->B.u_anchor_pos.size() + r->B.p_clipped_index.size();
    fighost::Results R;
    R.filled_len.assign(filled_len, filled_len + ng); R.gaptofill.assign(gaptofill, gaptofill + ng);
    R.str_off.assign(str_off, str_off + ng + 1);
    R.str.assign(str, str + (size_t)str_off[ng]);
    if (draw_pos && draw_isz && draw_len) {
        R.draw_pos.assign(draw_pos, draw_pos + nr); R.draw_isz.assign(draw_isz, draw_isz + nr); R.draw_len.assign(draw_len, draw_len + 2 * ng);
    }
    std::string e;
    if (!fighost::write_gapout(r->a, r->B, R, e) || !fighost::write_draw(r->a, r->B, R, e) || !fighost::write_gaploads(r->a, r->B, e) ||
        !fighost::write_scaffold(r->a, r->sc, r->B, R, e)) {
        set_err(err, errcap, e); return -1;
    }
    return 0;
}
