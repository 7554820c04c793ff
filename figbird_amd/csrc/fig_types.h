// fig_types.h -- POD descriptors shared by the host packer (fig_abi.cpp) and the gfx950
// gap-fill engine (fig_engine.h).  Everything here is plain data laid out for HBM:
//  * read bases are packed 2 bits/base (A0 C1 G2 T3) in 32-bit words, 16 bases per word,
//    followed by a 1 bit/base N-mask (32 bases per word); a read's words are contiguous so a
//    wavefront fetches a whole read with one coalesced load of <= 19 dwords (L <= 200);
//  * per-gap flank windows are byte codes (0..4) of the FW bases either side of the gap;
//  * per-gap mutable state lives in a per-workgroup scratch slab (see fig_scratch_layout).
#ifndef FIG_TYPES_H
#define FIG_TYPES_H
#include <stdint.h>

#define FIG_MAX_READLEN 200          // MAX_READLENGTH, Figbird.cpp:18
#define FIG_READ_CAP 3000            // partial_limit / unmapped_limit, Figbird.cpp:114-115
#define FIG_MAX_GAP 100000           // MAX_GAP, Figbird.cpp:30
#define FIG_FLANK 208                // flank window kept per side: >= max(read length, side_limit=30)
#define FIG_DBL_MAX 1.7976931348623157e308
#define FIG_SH_C 32                  // reads per chunk of the shared-factor E-step (fig_engine_shared.h)
#define FIG_SH_SC 4                  // chunks per super-chunk: the (chunk, tile) work items of phase A are dealt over the waves per super-chunk
#define FIG_SH_ONE 16                // operand-select register offset of the constant 1.0 (slots without a regular read)
#define FIG_MLE_FB 208               // doubles of the per-wave factor buffer of the MLE pass (>= FIG_MAX_READLEN)
// LDS-tiled class: doubles of LDS the MLE pass needs to run its LDS form over all ncolE columns: C[5][ncolE], one 16-byte
// packed-consensus record per column, the per-wave factor buffers, the packed consensus words (kc, kn) and slack
#define FIG_TILED_MLE_DOUBLES(ncolE, nw) (7LL * (ncolE) + (long long)(nw) * FIG_MLE_FB + ((ncolE) + 15) / 16 + 16 + 8)

struct FigDevModel {
    int32_t L;                       // maxReadLength
    int32_t Tmin, Tmax, cutoff;      // insertThresholdMin/Max, gapProbCutOff
    int32_t partial_flag, unmapped, script_itr, D, read_length, neg_overlap, partial_len, unm_limit;
    int32_t max_insert;
    double T[25];                    // errorTypeProbs[from][to]
    const double *e;                 // errorPosDist[k]
    const double *ome;               // pair tables: {1-e,e} fwd [2L], rev [2L]; {1-e-ins-del,e} fwd [2L], rev [2L]
    const double *ome1;              // 1 - errorPosDist[k]
    const double *m3;                // 1 - errorPosDist[k] - inPosDist[k] - delPosDist[k]
    const double *insd;              // insertLengthDistSmoothed[max_insert]
    const double *qtab;              // [256] pow(10, -(c-33)/10.0)   (qualityFilter, Figbird.cpp:1791-1792)
    double fmm_up;                   // upper bound of any MLE mismatch factor e[k]*T[from][to], from != to (k-mer prefilter)
};

struct FigDevGap {
    int64_t gapStart, contigLen;
    int32_t G0, stat2, stat3, fillflag;
    int32_t nU; int32_t nP;          // reads of this gap (nP = all lines of the partial file, <= 3001 kept)
    int64_t uBase, pBase;            // first read index in the batch-wide arrays
    int32_t alloc_arg; float gpf1, gpf2; int32_t lgf;
    int64_t flankOff;                // byte offset of this gap's 2*FIG_FLANK flank codes
    int64_t strOff;                  // offset of this gap's result string
    int32_t gapNo; int32_t cls;
    int64_t persistOff;              // this gap's persistent slab (candidate-parallel mode)
    int32_t capGg, rangeCap, nslots, pad;   // pad bit 0: a partial read of the gap holds a base outside ACGT
    int64_t streamOff;               // first dword of this gap's operand-select stream (FigDevBatch::ustream)
};

struct FigDevReads {                 // batch-wide SoA, one entry per read
    const int32_t *pos;              // unmapped: anchor pos ; partial: pos (col 4)
    const int32_t *aux;              // unmapped: isReverse  ; partial: match (col 3)
    const int32_t *clip;             // partial: clipped_index
    const int32_t *refpos;           // partial: mate pos or -1
    const int32_t *len;
    const int64_t *woff;             // word offset of the read's packed bases
    const int64_t *qoff;             // partial: byte offset of the quality string (or NULL)
};

struct FigDevBatch {
    int64_t n_gaps;
    const FigDevGap *gaps;
    const int32_t *order;            // gap indices, most expensive first
    FigDevReads u, p;
    const uint32_t *packed;          // all packed reads
    const uint32_t *ustream;         // operand-select stream of the unmapped reads: per gap [chunk of 32 reads][step j < L][32] 16-bit entries (fig_engine_shared.h)
    const uint8_t *qual;             // partial qualities
    const uint8_t *flank;            // per-gap flank codes
    // outputs
    int32_t *filled_len, *gaptofill; char *str;
    int32_t dbg_max_cand; int32_t *dbg_n_cand; int32_t *dbg_cand_i; double *dbg_cand_lik; int32_t *dbg_n_place;
    int32_t *draw_pos, *draw_isz, *draw_len; int64_t n_ureads;
    int32_t dbg_plane_cols, dbg_plane_reads; double *dbg_counts, *dbg_read_maxlv;   // numeric planes (i)/(ii), parity tests only
    // work queue + counters
    int32_t *queue_head;             // [1] next index into order[]
    unsigned long long *counters;    // [0] placeReads calls, [1] algorithmic flops (as integer count)
    // scratch
    uint8_t *scratch; int64_t scratch_stride;   // one slab per workgroup
    int32_t capG, capR, capP, capC;  // capacities the slab was carved for (columns, unmapped reads, partial reads, candidates)
    int32_t capW, capE;              // weight-buffer doubles, extended-table columns
    uint8_t *persist;                // per-gap persistent slabs
    int32_t *gapctl;                 // [n_gaps*4] {status (0 finished, 1 more candidates, 2 loop done), next j, range, reach bit of fig_probe_kernel}
    const uint8_t *ot_preset;        // [n_gaps] 1 = the gap's worker process has set overlap_threshold before it gets to the gap (Figbird.cpp:103, :6317)
};

#endif
