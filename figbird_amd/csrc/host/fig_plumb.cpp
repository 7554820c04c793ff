// fig_plumb.cpp -- the scaffold / report plumbing either side of the fill, folded into the C++ host:
//   flank_trim   = FlankTrim.cpp:21-233      (RunFigbird.sh:254,433: iteration 1 only)
//   rewrap_fasta = reference.py:1-29         (RunFigbird.sh:256,435,809)
//   reduce_scf   = Reduce_SCF.cpp:14-152     (RunFigbird.sh:266,320)
//   combine_gaps = CombineGaps.cpp:169-313   (RunFigbird.sh:777)
// Same files in, byte-identical files out (tests/test_plumbing.py compares with the reference's own binaries, compiled
// where they lie into oracle/_ref/, and with committed goldens).  Quirks are kept on purpose and marked "quirk".
#include "fig_plumb.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace figplumb {

// ---- the reference's FASTA reader (FlankTrim.cpp:70-134, Reduce_SCF.cpp:59-135, Preprocess.cpp:2017-2084): fgets with a
// 1024-byte buffer; a piece that fills the buffer (1023 chars, no newline) is kept whole, any shorter piece loses its last
// character -- the newline, or a base when the file's last line has no newline (quirk); lines starting with ';' are skipped.
static const int kRec = 1024;

// visit(header_index_or_-1, piece): pieces of sequence in file order; `on_header(i)` fires for each header line
template <class OnHeader, class OnPiece>
static bool scan_fasta(const std::string &path, OnHeader on_header, OnPiece on_piece) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    std::vector<char> line(kRec);
    while (fgets(line.data(), kRec, f) != nullptr) {
        if (line[0] == ';') continue;
        size_t n = strlen(line.data());
        if (line[0] == '>') { on_header(line.data(), n); continue; }
        if ((long)n < kRec - 1) on_piece(line.data(), n - 1);       // drop the last character
        else on_piece(line.data(), n);                              // a full buffer: 1023 characters kept
    }
    fclose(f);
    return true;
}

// ---- FlankTrim -------------------------------------------------------------------------------------------------------
bool flank_trim(const std::string &in, int trimsize, int readlen, const std::string &out, std::string &err) {
    std::vector<std::string> names, contigs;
    std::string cur;
    bool ok = scan_fasta(in,
        [&](const char *l, size_t n) {
            // contigNames gets EVERY header's first token (FlankTrim.cpp:79-82); the contig before it is pushed only when it is
            // non-empty (:84-95) -- so an empty record shifts names against sequences (quirk, kept)
            std::string nm(l + 1, n >= 2 ? n - 2 : 0);
            size_t e = nm.find_first_of(" \t\n");
            std::string tok = e == std::string::npos ? nm : nm.substr(0, e);
            if (tok.empty()) { size_t b = nm.find_first_not_of(" \t\n"); if (b != std::string::npos) { size_t e2 = nm.find_first_of(" \t\n", b); tok = nm.substr(b, e2 == std::string::npos ? std::string::npos : e2 - b); } }
            names.push_back(tok.empty() ? "(null)" : tok);       // strtok() skips leading delimiters; NULL prints as "(null)"
            if (!cur.empty()) { contigs.push_back(cur); cur.clear(); }
        },
        [&](const char *l, size_t n) { cur.append(l, n); });
    if (!ok) { err = "Can't open gapped genome file"; return false; }
    contigs.push_back(cur);                                          // the last contig is pushed unconditionally (:136-139)
    FILE *fo = fopen(out.c_str(), "w");
    if (!fo) { err = "can't write " + out; return false; }
    int nStart = 0;
    unsigned long nStartPos = 0, nCount = 0;
    for (size_t i = 0; i < contigs.size(); i++) {
        std::string &c = contigs[i];
        const unsigned long clen = c.size();
        fprintf(fo, ">%s\n", i < names.size() ? names[i].c_str() : "");
        for (long j = 0; j < (long)clen; j++) {
            const bool isN = c[j] == 'N' || c[j] == 'n';
            if (isN) {
                if (nStart == 0) { nStart = 1; nCount = 1; nStartPos = (unsigned long)j; }
                else nCount++;
            }
            if ((!isN && nStart == 1) || (isN && j == (long)clen - 1)) {
                // gaps of 2..readlen-1 N with more than 2*trim clean bases beyond the trimmed flank either side (:175-176;
                // the right-hand test is unsigned arithmetic, kept)
                if (trimsize > 0 && nCount > 1 && (int)nCount < readlen && (int)nStartPos - trimsize > 2 * trimsize &&
                    (unsigned long)(clen - nStartPos - nCount) > (unsigned long)(2 * trimsize)) {
                    bool clean = true;
                    for (int t = 0; t < trimsize; t++) if (c[nStartPos - t - 1] == 'N' || c[nStartPos + nCount + t] == 'N') clean = false;   // strpbrk(..., "N"): upper case only
                    if (clean) {
                        for (int t = 0; t < trimsize; t++) { c[nStartPos - t - 1] = 'N'; c[nStartPos + nCount + t] = 'N'; }
                        j += trimsize;
                    }
                }
                nStart = 0;
            }
        }
        fprintf(fo, "%s\n", c.c_str());
    }
    fclose(fo);
    return true;
}

// ---- reference.py ----------------------------------------------------------------------------------------------------
bool rewrap_fasta(const std::string &in, const std::string &out, int slice, std::string &err) {
    FILE *fi = fopen(in.c_str(), "r");
    if (!fi) { err = "can't open " + in; return false; }
    FILE *fo = fopen(out.c_str(), "w");
    if (!fo) { fclose(fi); err = "can't write " + out; return false; }
    if (slice <= 0) { fclose(fi); fclose(fo); err = "slice must be positive"; return false; }
    std::string data;
    { char buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, fi)) > 0) data.append(buf, n); }
    fclose(fi);
    size_t saved = 0;                                   // bufferlen_saved of the previous line
    size_t p = 0;
    while (p < data.size()) {
        size_t e = data.find('\n', p);
        const bool has_nl = e != std::string::npos;
        std::string line = data.substr(p, has_nl ? e - p + 1 : std::string::npos);
        p = has_nl ? e + 1 : data.size();
        std::string buffer;
        if (line[0] == '>') {
            if (saved > 0) fputc('\n', fo);             // buffer is '' here: a lone newline closes the previous sequence (reference.py:11-13)
            fwrite(line.data(), 1, line.size(), fo);
        } else buffer = line.substr(0, line.size() - 1); // line[:-1]: drops the newline -- or a base on an unterminated last line (quirk)
        size_t blen = buffer.size();
        saved = blen;
        size_t i = 0;
        while (true) {
            if (blen >= (size_t)slice) { fwrite(buffer.data() + slice * i, 1, (size_t)slice, fo); fputc('\n', fo); blen -= (size_t)slice; i++; }
            else { if (saved > (size_t)slice * i) fwrite(buffer.data() + slice * i, 1, saved - (size_t)slice * i, fo); break; }
        }
    }
    fclose(fo);
    return true;
}

// ---- Reduce_SCF ------------------------------------------------------------------------------------------------------
bool reduce_scf(const std::string &in, const std::string &tmp_dir, std::string &err) {
    FILE *fo = nullptr;
    std::string cur, name;
    bool have_name = false;
    int Nflag = 0;
    // the output is opened after the input (Reduce_SCF.cpp:24-36): a missing input leaves no newgenome.fa behind
    {
        FILE *probe = fopen(in.c_str(), "r");
        if (!probe) { err = "Can't open gapped genome file during reduction"; return false; }
        fclose(probe);
    }
    fo = fopen((tmp_dir + "newgenome.fa").c_str(), "w");
    if (!fo) { err = "can't write " + tmp_dir + "newgenome.fa"; return false; }
    auto flush = [&]() {
        if (Nflag == 1) { Nflag = 0; fprintf(fo, ">%s\n%s\n", have_name ? name.c_str() : "(null)", cur.c_str()); }
    };
    scan_fasta(in,
        [&](const char *l, size_t n) {
            if (!cur.empty()) { flush(); cur.clear(); }          // (an empty record keeps its Nflag for the next one: it has none)
            name.assign(l + 1, n >= 2 ? n - 2 : 0);              // the WHOLE header line minus its last character (:83-85), not the first token
            have_name = true;
        },
        [&](const char *l, size_t n) {
            // the N test looks at the raw piece including the character that is dropped afterwards (:91-100)
            if (Nflag == 0) { size_t raw = (long)n < kRec - 1 ? n + 1 : n; for (size_t z = 0; z < raw; z++) if (l[z] == 'N' || l[z] == 'n') { Nflag = 1; break; } }
            cur.append(l, n);
        });
    flush();
    fclose(fo);
    return true;
}

// ---- CombineGaps -----------------------------------------------------------------------------------------------------
namespace {
struct CGap { std::string s; bool has_s = false; int left_start_N = -1, right_end_N = -1, fully_closed = 0, originalGap = 0, finalGapLen = 0, r_size = 0; };
int g_pos[3] = {0, 0, 0};                                  // `pos` is a global in the reference and keeps stale values (CombineGaps.cpp:15)
int check_complete(const std::string &gapstr) {           // :32-63
    int Nstart = 0, region_count = 0, Ncount = 0;
    const int len = (int)gapstr.size();
    for (int i = 0; i < len; i++) {
        if (gapstr[i] == 'N' && Nstart == 0) { Nstart = 1; g_pos[0] = i; Ncount++; }
        else if (gapstr[i] != 'N' && Nstart == 1) { Nstart = 0; region_count++; g_pos[1] = i - 1; g_pos[2] = Ncount; }
        if (i == len - 1 && Nstart == 1) { region_count++; g_pos[1] = i; g_pos[2] = Ncount; }
    }
    return region_count;
}
void combine(CGap &g, int org, int gaplen, const std::string &s, int rc, int itr) {     // :65-124
    if (itr == 1) g.originalGap = org;
    g.fully_closed = 1 - rc;
    if (gaplen == 0) { g.s.clear(); g.has_s = true; g.finalGapLen = 0; return; }
    if (itr == 1) {
        g.s = s; g.has_s = true; g.finalGapLen = gaplen;
        if (g.fully_closed != 1) { check_complete(g.s); g.left_start_N = g_pos[0]; g.right_end_N = g_pos[1]; g.r_size = gaplen - g.right_end_N; }
    } else {
        const int newlen = g.left_start_N + gaplen + g.r_size;
        std::string ns;
        for (int i = 0; i < g.left_start_N; i++) ns.push_back(i < (int)g.s.size() ? g.s[i] : '\0');
        for (int i = 0; i < gaplen; i++) ns.push_back(i < (int)s.size() ? s[i] : '\0');
        for (int i = 0; i < g.r_size - 1; i++) { int k = i + 1 + g.right_end_N; ns.push_back(k >= 0 && k < (int)g.s.size() ? g.s[k] : '\0'); }
        if ((int)ns.size() > newlen - 1 && newlen >= 1) ns.resize((size_t)newlen - 1);      // newgap_str[newlen-1] = '\0'
        size_t z = ns.find('\0'); if (z != std::string::npos) ns.resize(z);                // strcpy stops at the first NUL
        g.s = ns;
        check_complete(g.s);
        g.left_start_N = g_pos[0]; g.right_end_N = g_pos[1];
        g.finalGapLen = newlen - 1;
        g.r_size = g.finalGapLen - g.right_end_N;
    }
}
// fscanf(f, "%d\t%d\t%d\t%d\t%d\t", ...) then "%s\n": whitespace-separated tokens
bool next_token(FILE *f, std::string &tok) {
    int c;
    while ((c = fgetc(f)) != EOF && (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f')) {}
    if (c == EOF) return false;
    tok.clear();
    do { tok.push_back((char)c); c = fgetc(f); } while (c != EOF && !(c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'));
    if (c != EOF) ungetc(c, f);
    return true;
}
}  // namespace

int combine_gaps(int num_itr, const std::string &path, std::string &err) {
    std::vector<CGap> gaps;
    int totalgaps = 0;
    g_pos[0] = g_pos[1] = g_pos[2] = 0;
    for (int itr = 1; itr <= num_itr; itr++) {
        const std::string fn = path + "gapout_" + std::to_string(itr) + ".txt";
        FILE *f = fopen(fn.c_str(), "r");
        if (!f) { err = "Can't open gapout txt file"; return 1; }
        if (itr == 1) {
            // one gap per fgets() piece of at most 10023 characters (:201-204): a gap string longer than that counts twice (quirk, kept)
            std::vector<char> line(10024);
            while (fgets(line.data(), 10024, f) != nullptr) totalgaps++;
            fclose(f);
            gaps.assign((size_t)totalgaps, CGap());
            f = fopen(fn.c_str(), "r");
            if (!f) { err = "Can't open gapout txt file"; return 1; }
        }
        for (int g = 0; g < totalgaps; g++) {
            if (gaps[g].fully_closed != 0) continue;
            std::string tok;
            int vals[5] = {0, 0, 0, 0, 0};
            for (int k = 0; k < 5; k++) { if (!next_token(f, tok)) break; vals[k] = atoi(tok.c_str()); }
            const int gapLength = vals[3], gapStringLength = vals[4];
            if (gapStringLength > 0) {
                std::string s;
                next_token(f, s);
                const int rc = check_complete(s);
                if (rc > 1) { fclose(f); return 0; }              // exit(0) without a word (:252-256): outputs of the previous iteration stay
                combine(gaps[g], gapLength, gapStringLength, s, rc, itr);
            } else combine(gaps[g], gapLength, gapStringLength, std::string(), 0, 1);
        }
        fclose(f);
        FILE *fo = fopen((path + "combined_gapstring.txt").c_str(), "w");
        if (!fo) { err = "can't write combined_gapstring.txt"; return 1; }
        for (int g = 0; g < totalgaps; g++) fprintf(fo, "%s\n", gaps[g].has_s ? gaps[g].s.c_str() : "(null)");
        fclose(fo);
    }
    FILE *fo = fopen((path + "Individual_gaps.txt").c_str(), "w");
    FILE *fi = fopen((path + "combined_gapstring.txt").c_str(), "r");
    if (!fo || !fi) { if (fo) fclose(fo); if (fi) fclose(fi); err = "can't write Individual_gaps.txt"; return 1; }
    fprintf(fo, "GapNo\tOriginal_Length\tFilled_Length\n\n");
    std::string line;
    for (int g = 0; g < totalgaps; g++) {
        if (gaps[g].finalGapLen > 0) { std::string t; if (next_token(fi, t)) line = t; }     // a failed fscanf leaves `line` as it was
        else line.clear();
        fprintf(fo, "%d\t%d\t%d\t%s\n", g, gaps[g].originalGap, gaps[g].finalGapLen, line.c_str());
    }
    fclose(fo); fclose(fi);
    return 0;
}

}  // namespace figplumb
