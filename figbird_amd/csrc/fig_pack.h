// fig_pack.h -- host-side packer: fig_gap_batch (caller arrays) -> device-layout arrays.
// Pure C++ (no HIP) so that the CPU unit-test emulation (tests/emu) packs exactly like
// libfighip.so does.  Role in the reference: the per-gap preamble of Figbird.cpp's main loop
// (findFrac/alloc_arg :6879-6906,:7393-7400; read caps :5763,:1814; flank window :2342-2372).
#ifndef FIG_PACK_H
#define FIG_PACK_H
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/figbird_hip.h"
#include "fig_types.h"
#include "fig_gaprules.h"
// (fig_engine.h must already be included: fig_persist_layout / FigState sizes come from it)

static inline int code_of(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

// Packed read = [2-bit words: ceil(len/16)] [N-mask words: ceil(len/32)] [5 per-base counts, 2 words]
//               [positions of A, then C, G, T, N bases, each list in DESCENDING order: ceil(len/4) words].
// The position lists let the column-accumulation pass of the E-step run one branch-free add chain per base.
static void pack_read(const char *s, int len, std::vector<uint32_t> &out) {
    // record: 2-bit codes | N mask | {cntA,cntC,cntG,cntT},{cntN} | per-base position lists (descending j), each
    // list padded to a dword boundary so that the column pass can take four positions from one register
    int nw2 = (len + 15) >> 4, nwm = (len + 31) >> 5, nwp = ((len + 3) >> 2) + 4;
    size_t base = out.size();
    out.resize(base + nw2 + nwm + 2 + nwp, 0u);
    int cnt[5] = {0, 0, 0, 0, 0};
    for (int j = 0; j < len; j++) {
        int c = code_of(s[j]);
        if (c < 4) out[base + (j >> 4)] |= (uint32_t)c << ((j & 15) * 2);
        else out[base + nw2 + (j >> 5)] |= 1u << (j & 31);
        cnt[c]++;
    }
    uint8_t *cb = (uint8_t *)&out[base + nw2 + nwm];
    for (int c = 0; c < 5; c++) cb[c] = (uint8_t)cnt[c];
    uint8_t *pl = (uint8_t *)&out[base + nw2 + nwm + 2];
    int k = 0;
    for (int c = 0; c < 5; c++) {
        for (int j = len - 1; j >= 0; j--) if (code_of(s[j]) == c) pl[k++] = (uint8_t)j;
        k = (k + 3) & ~3;
    }
}

// Upper bound of the MLE pass's mismatch factors e[k] * errorTypeProbs[from][to] (from != to; from = 4 is an N column):
// what the k-mer prefilter of fig_hot_mle raises to the number of mismatches.  Rounded up a little so that products
// of it stay upper bounds in floating point.
static inline double fig_model_fmm(const fig_model *m) {
    double emax = 0, tmax = 0;
    for (int k = 0; k < m->max_read_length; k++) emax = std::max(emax, m->error_pos_dist[k]);
    for (int from = 0; from < 5; from++) for (int to = 0; to < 4; to++) if (from != to) tmax = std::max(tmax, m->error_type_probs[from * 5 + to]);
    // every exact-pruning argument of the MLE pass needs factors in [0, 1]; a model outside that (possible only with
    // degenerate training data) is marked with 2.0 and the engine then evaluates every placement in full
    for (int k = 0; k < m->max_read_length; k++) {
        const double e = m->error_pos_dist[k], m3 = 1 - e - m->in_pos_dist[k] - m->del_pos_dist[k];
        if (!(e >= 0 && e <= 1 && m3 >= 0 && m3 <= 1)) return 2.0;
    }
    for (int i = 0; i < 25; i++) if (!(m->error_type_probs[i] >= 0 && m->error_type_probs[i] <= 1)) return 2.0;
    double f = emax * tmax * (1.0 + 1e-9);
    if (!(f > 0)) f = 1.0;
    return f < 1.0 ? f : 1.0;
}

#define FIG_SPLIT_MIN_GAPS 96        // a class with at least this many gaps is scheduled as two lanes
struct FigLaunchClass { int capG, capGl, ncolE, Wcap, nteams, nt; bool lds_tab; size_t lds; int q_begin, q_end;
                        int tiles, tile_step, tile_cols, tiled_max;
                        int nsplit; };     // tiled_max: doubles of LDS the table/weight area may grow to      // tiles > 0: table in HBM/L2, streamed through an LDS tile per chunk (fig_hot_estep)

struct FigPacked {
    std::vector<FigDevGap> gaps;
    std::vector<int32_t> order;
    std::vector<uint8_t> flank, qual;
    std::vector<uint32_t> packed;
    std::vector<int32_t> u_pos, u_aux, u_len, p_pos, p_aux, p_clip, p_ref, p_len;
    std::vector<int64_t> u_woff, p_woff, p_qoff, str_off;
    std::vector<FigLaunchClass> classes;
    std::vector<uint8_t> ot_preset;   // per gap: overlap_threshold already set by an earlier gap of its worker process (fig_gap_batch::gap_ot_preset)
    bool ot_given = false;            // the caller supplied it; otherwise the library measures it (one process, batch order)
    int capR = 1, capP = 1, capC = 1, capG = 8, capW = 0, capE = 0;
    int64_t str_total = 0, n_gaps = 0, persist_total = 0, stream_total = 0;   // stream_total: dwords of the operand-select stream
    int nslots = 64;                  // speculative candidate slots per gap (fig_engine_sched.h)
    int64_t packed_bytes() const {
        return (int64_t)(packed.size() * 4 + flank.size() + gaps.size() * sizeof(FigDevGap) + qual.size() +
                         (u_pos.size() * 3 + p_pos.size() * 5) * 4 + (u_woff.size() + p_woff.size() + p_qoff.size()) * 8);
    }
};

// lds_fixed = bytes of LDS the kernel needs besides P/Q and the weight buffer, as a function of capG.
static int fig_pack(const fig_model *m, const fig_gap_batch *b, size_t state_bytes, FigPacked &K) {
    int64_t ng = b->n_gaps;
    if (ng < 0) return FIG_EINVAL;
    if (ng > 0 && (!b->gap_contig || !b->gap_start || !b->gap_len || !b->contig_off || !b->contig_seq || !b->p_read_off)) return FIG_EINVAL;
    if (m->unmapped_flag && ng > 0 && !b->u_read_off) return FIG_EINVAL;
    // CSR ranges must be non-negative and non-decreasing, and the arrays behind a non-empty range present
    if (ng > 0) {
        if (b->p_read_off[0] < 0 || (m->unmapped_flag && b->u_read_off[0] < 0)) return FIG_EINVAL;
        for (int64_t g = 0; g < ng; g++) {
            if (b->p_read_off[g + 1] < b->p_read_off[g]) return FIG_EINVAL;
            if (m->unmapped_flag && b->u_read_off[g + 1] < b->u_read_off[g]) return FIG_EINVAL;
        }
        if (b->p_read_off[ng] > 0 && (!b->p_seq_off || !b->p_seq || !b->p_pos || !b->p_match || !b->p_clipped_index || !b->p_ref_pos)) return FIG_EINVAL;
        if (m->unmapped_flag && b->u_read_off[ng] > 0 && (!b->u_seq_off || !b->u_seq || !b->u_anchor_pos || !b->u_is_reverse)) return FIG_EINVAL;
    }
    if (m->max_distance < m->max_read_length) return FIG_EUNSUP;       // placements must stay inside the +-D window
    K.n_gaps = ng;
    K.gaps.assign(ng, FigDevGap());
    K.flank.assign((size_t)ng * 2 * FIG_FLANK, 4);
    K.str_off.assign(ng + 1, 0);
    std::vector<double> cost(ng, 0.0);
    int64_t str_total = 0;
    for (int64_t g = 0; g < ng; g++) {
        FigDevGap &d = K.gaps[g];
        int ci = b->gap_contig[g];
        if (ci < 0 || ci >= b->n_contigs) return FIG_EINVAL;
        int64_t c0 = b->contig_off[ci], c1 = b->contig_off[ci + 1];
        d.contigLen = c1 - c0; d.gapStart = b->gap_start[g]; d.G0 = b->gap_len[g];
        if (d.G0 < 1 || d.gapStart < 0 || d.gapStart + d.G0 > d.contigLen) return FIG_EINVAL;
        d.stat2 = b->gap_stat2 ? b->gap_stat2[g * 3 + 1] : 0;
        d.stat3 = b->gap_stat2 ? b->gap_stat2[g * 3 + 2] : 0;
        d.fillflag = b->gap_fillflag ? b->gap_fillflag[g] : 1;
        gap_alloc(m, d.G0, &d.alloc_arg, &d.gpf1, &d.gpf2, &d.lgf);
        if (d.alloc_arg < d.G0) d.alloc_arg = d.G0;
        d.gapNo = (int32_t)g;
        d.flankOff = g * 2 * FIG_FLANK;
        uint8_t *fl = &K.flank[(size_t)d.flankOff];
        const char *cs = b->contig_seq + c0;
        for (int k = 1; k <= FIG_FLANK; k++) { int64_t a = d.gapStart - k; fl[k - 1] = a >= 0 ? (uint8_t)code_of(cs[a]) : 4; }
        for (int k = 0; k < FIG_FLANK; k++) { int64_t a = d.gapStart + d.G0 + k; fl[FIG_FLANK + k] = a < d.contigLen ? (uint8_t)code_of(cs[a]) : 4; }
        d.strOff = str_total; K.str_off[g] = str_total; str_total += d.alloc_arg + 1;
        d.uBase = (int64_t)K.u_pos.size(); d.nU = 0;
        if (m->unmapped_flag) {
            int64_t r0 = b->u_read_off[g], r1 = b->u_read_off[g + 1];
            int64_t n = r1 - r0;
            if (n > FIG_READ_CAP) n = FIG_READ_CAP;                      // parseUnmapped stops at unmapped_limit (:5763)
            for (int64_t r = r0; r < r0 + n; r++) {
                int len = (int)(b->u_seq_off[r + 1] - b->u_seq_off[r]);
                if (len > FIG_MAX_READLEN || len > m->max_read_length || len < 1) return FIG_EUNSUP;
                K.u_pos.push_back(b->u_anchor_pos[r]);
                {   // aux: bit 0 = mate was reverse-complemented, bit 1 = the read has a base outside ACGT (generic chain)
                    int hasN = 0;
                    const char *rs_ = b->u_seq + b->u_seq_off[r];
                    for (int q = 0; q < len; q++) if (code_of(rs_[q]) > 3) { hasN = 1; break; }
                    K.u_aux.push_back((b->u_is_reverse[r] ? 1 : 0) | (hasN << 1));
                }
                K.u_len.push_back(len);
                K.u_woff.push_back((int64_t)K.packed.size());
                pack_read(b->u_seq + b->u_seq_off[r], len, K.packed);
            }
            d.nU = (int32_t)n;
        }
        d.pBase = (int64_t)K.p_pos.size(); d.nP = 0;
        {
            int64_t r0 = b->p_read_off[g], r1 = b->p_read_off[g + 1];
            int64_t n = r1 - r0;
            if (n > FIG_READ_CAP + 1) n = FIG_READ_CAP + 1;              // every per-file loop stops after 3001 lines
            for (int64_t r = r0; r < r0 + n; r++) {
                int len = (int)(b->p_seq_off[r + 1] - b->p_seq_off[r]);
                if (len > FIG_MAX_READLEN || len > m->max_read_length || len < 1) return FIG_EUNSUP;   // the {1-e,e} pair tables hold max_read_length entries
                { const char *rs_ = b->p_seq + b->p_seq_off[r]; for (int q = 0; q < len; q++) if (code_of(rs_[q]) > 3) { d.pad |= 1; break; } }   // bit 0: the gap has a partial read with a base outside ACGT (fig_engine_partial.h)
                K.p_pos.push_back(b->p_pos[r]); K.p_aux.push_back(b->p_match[r]); K.p_clip.push_back(b->p_clipped_index[r]);
                K.p_ref.push_back(b->p_ref_pos[r]); K.p_len.push_back(len);
                K.p_woff.push_back((int64_t)K.packed.size());
                pack_read(b->p_seq + b->p_seq_off[r], len, K.packed);
                K.p_qoff.push_back((int64_t)K.qual.size());
                if (b->p_qual) K.qual.insert(K.qual.end(), b->p_qual + b->p_seq_off[r], b->p_qual + b->p_seq_off[r] + len);
                else K.qual.insert(K.qual.end(), (size_t)len, (uint8_t)'I');
            }
            d.nP = (int32_t)n;
        }
        K.capR = std::max(K.capR, d.nU); K.capP = std::max(K.capP, d.nP);
        int range = gap_range(d.G0, d.gpf1, d.gpf2);
        K.capC = std::max(K.capC, range);
        K.capG = std::max(K.capG, d.alloc_arg);
        double L = m->max_read_length;
        if (m->unmapped_flag) cost[g] = (double)d.nU * std::min<double>(d.alloc_arg + L, 2200.0) * L * (d.lgf ? 2.0 : range * (d.G0 <= m->unm_limit / 3 ? 10.5 : 11.5));      // placeReads calls: measured medians (tools/cost_model_check.py)
        else cost[g] = (double)d.nP * L * L * range * 3.0;
        if (d.fillflag == -1) cost[g] = 1;
    }
    K.str_total = str_total; K.str_off[ng] = str_total;
    {   // operand-select stream of the shared-factor E-step: per gap ceil(nU/32) chunks x L steps x 16 dwords (32 16-bit entries), built on the device
        int64_t so = 0;
        for (int64_t g = 0; g < ng; g++) {
            K.gaps[g].streamOff = so;
            so += (int64_t)((K.gaps[g].nU + FIG_SH_C - 1) / FIG_SH_C) * m->max_read_length * (FIG_SH_C / 2);
        }
        K.stream_total = so + FIG_SH_C;        // + one padded step: the chain prefetches the row after the last one
    }
    K.capG = (K.capG + 7) & ~7;
    {   // which gaps find the reference's process-level overlap_threshold already at 5 (Figbird.cpp:103, :6317): given by the
        // caller (gap_ot_preset: it knows which worker process of the reference a gap would run in, and which gaps of that
        // process are in other shards), else MEASURED by the library after the upload (fig_abi.hip: probe_and_preset) for ONE
        // process taking the gaps in batch order (numthreads = 1)
        K.ot_given = b->gap_ot_preset != nullptr;
        K.ot_preset.assign((size_t)std::max<int64_t>(ng, 1), 0);
        for (int64_t g = 0; g < ng; g++) if (b->gap_ot_preset) K.ot_preset[(size_t)g] = b->gap_ot_preset[g] ? 1 : 0;
    }
    // ---- launch classes by the longest candidate a gap can reach (LDS columns); within a class the most
    // expensive gaps come first.  gmax: candidate range (:6237-6238), checkGapReads probes (:6121-6153).
    std::vector<int> gmax(ng, 8);
    for (int64_t g = 0; g < ng; g++) {
        const FigDevGap &d = K.gaps[g];
        int gm = std::max(d.G0, (int)(d.G0 * d.gpf2));
        if (m->unmapped_flag && d.G0 <= m->unm_limit) gm = std::max(gm, d.G0 < 30 ? 70 : 3 * d.G0);
        gmax[g] = std::min(gm, d.alloc_arg);
    }
    // per-gap persistent slabs of the candidate-parallel scheduler
    {
        long long off = 0;
        for (int64_t g = 0; g < ng; g++) {
            FigDevGap &d = K.gaps[g];
            d.capGg = (gmax[g] + 7) & ~7; d.rangeCap = gap_range(d.G0, d.gpf1, d.gpf2); d.nslots = K.nslots;
            d.persistOff = off;
            off += (fig_persist_layout(nullptr, d.capGg, d.nU, d.nP, d.rangeCap, d.nslots, state_bytes, nullptr) + 255) & ~255LL;
        }
        K.persist_total = off;
    }
    struct ClsDef { int capGl, nt; };
    const ClsDef defs[] = {{448, 256}, {1216, 512}, {1600, 512}, {1 << 30, 512}};   // longest candidate (columns), threads
    K.order.clear(); K.classes.clear();
    const int L1 = m->max_read_length - 1;
    int prevcap = 0;
    for (const ClsDef &cd : defs) {
        std::vector<int32_t> ids;
        int mx = 8, mxa = 8;
        for (int64_t g = 0; g < ng; g++)
            if (gmax[g] > prevcap && gmax[g] <= cd.capGl) { ids.push_back((int32_t)g); mx = std::max(mx, gmax[g]); mxa = std::max(mxa, K.gaps[g].alloc_arg); }
        prevcap = cd.capGl;
        if (ids.empty()) continue;
        std::stable_sort(ids.begin(), ids.end(), [&](int32_t a, int32_t c) { return cost[a] > cost[c]; });
        FigLaunchClass c;
        c.capG = (mxa + 7) & ~7;
        c.capGl = (mx + 7) & ~7;
        c.ncolE = (c.capGl + 2 * L1 + 7) & ~7;
        c.Wcap = (c.capGl + m->max_read_length + 7) & ~7;
        c.nt = cd.nt;
        const size_t LDS_MAX = 160 * 1024 - 1024;
        size_t fixed = state_bytes + (size_t)c.capGl + FIG_MAX_READLEN + 64 + FIG_PLB_BYTES;
        int nw = c.nt / 64;
        c.lds_tab = false; c.nteams = nw;
        for (int nt_ = nw; nt_ >= 1; nt_ >>= 1) {
            size_t need = fixed + sizeof(double) * ((size_t)9 * c.ncolE + (size_t)nt_ * c.Wcap);
            if (need <= LDS_MAX) { c.lds_tab = true; c.nteams = nt_; c.lds = need; break; }
        }
        c.tiles = 0; c.tile_step = 0; c.tile_cols = 0; c.tiled_max = 0;
        if (!c.lds_tab) {
            c.lds = fixed + 64; c.nt = 512; c.nteams = 8;
            // LDS-tiled form: the weight rows stay in LDS and the E-step streams the table through one LDS tile of columns at a
            // time (tiles overlap by a read length: a placement belongs to the tile that holds its first column).  Most rows first,
            // then fewest tiles.
            bool found = false;
            for (int nt_ = 4; nt_ >= 1 && !found; nt_ >>= 1)
                for (int ntl = 2; ntl <= 8 && !found; ntl++) {
                    int step = ((c.ncolE + ntl - 1) / ntl + 7) & ~7;
                    int tcols = (step + m->max_read_length + 8 + 7) & ~7;
                    size_t need = fixed + sizeof(double) * ((size_t)9 * tcols + (size_t)nt_ * c.Wcap);
                    if (need <= LDS_MAX) { found = true; c.tiles = ntl; c.tile_step = step; c.tile_cols = tcols; c.nteams = nt_; c.lds = need; }
                }
            if (found) {
                // the MLE pass of this class runs its LDS form when {C[5], packed-consensus records} of all columns + its
                // buffers fit (fig_tiled_mle_doubles); the area in front of FigState is the larger of the two layouts
                c.tiled_max = (int)((LDS_MAX - fixed) / sizeof(double));
                const long long mle = FIG_TILED_MLE_DOUBLES(c.ncolE, c.nt / 64);
                if (mle <= c.tiled_max && fixed + sizeof(double) * (size_t)mle > c.lds) c.lds = fixed + sizeof(double) * (size_t)mle;
            }
        }
        // A class with many gaps runs as TWO lanes of the same memory form (gaps dealt alternately in cost order), each with
        // its own stream, queue and round sequence: one lane's persistent workgroups take over the CUs the other lane's
        // round leaves idle while its last items finish (fig_abi.hip: run_class_parallel; nsplit halves the items per round).
        { const char *ev = getenv("FIG_CLASS_LANES");                                   // tuning knob: lanes per large class (default 2)
          const int want = ev ? std::max(1, std::min(8, atoi(ev))) : 2;
          c.nsplit = (int)ids.size() >= (FIG_SPLIT_MIN_GAPS / 2) * want ? want : 1; }
        for (int part = 0; part < c.nsplit; part++) {
            c.q_begin = (int)K.order.size();
            for (size_t k = (size_t)part; k < ids.size(); k += (size_t)c.nsplit) { const int32_t id = ids[k]; K.gaps[id].cls = (int)K.classes.size(); K.order.push_back(id); }
            c.q_end = (int)K.order.size();
            K.classes.push_back(c);
        }
        K.capE = std::max(K.capE, c.ncolE);
        // + slack: the column pass reads up to 2 tiles past a row's end (tiled class: the MLE pass keeps its buffers in the HBM rows)
        K.capW = std::max(K.capW, (c.tiles ? 8 : c.nteams) * c.Wcap + 1024);
        // product rows of the shared-factor E-step (fig_engine_shared.h): one super-chunk of reads x all placements
        if (m->unmapped_flag && c.lds_tab && !c.tiles && c.nteams >= 4) K.capW = std::max(K.capW, FIG_SH_SC * FIG_SH_C * c.Wcap + 1024);
    }
    return FIG_OK;
}

static int64_t fig_pack_results_capacity(const fig_model *m, const fig_gap_batch *b) {
    int64_t tot = 0;
    for (int64_t g = 0; g < b->n_gaps; g++) {
        int a, lgf; float f1, f2;
        gap_alloc(m, b->gap_len[g], &a, &f1, &f2, &lgf);
        tot += std::max(a, b->gap_len[g]) + 1;
    }
    return tot;
}
#endif
