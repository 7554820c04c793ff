/* figbird_hip.h -- C ABI of libfighip.so, the MI355X (gfx950) gap-fill engine.
 *
 * Drop-in boundary.  The reference has no in-process API: its
 * dispatcher FillGaps.cpp shells out `g++ Figbird.cpp && ./aN.out <16 args>` once per
 * thread (FillGaps.cpp:51-138) and each such process runs the per-gap loop
 * (Figbird.cpp:7329-7440) -> GapFiller::fillGap (Figbird.cpp:6201-6570).  This ABI
 * replaces exactly that: "fill this batch of gaps with this model", as plain C structs of
 * caller-owned host arrays.  No torch types, no C++ types, no global state.
 *
 * Conventions: every function returns 0 (FIG_OK) or a negative FIG_E* code and never
 * exits/throws; all sizes are int64_t; the caller owns every buffer; fig_fill_gaps() is
 * synchronous (the library syncs its own HIP stream before returning); one fig_ctx per
 * GPU, not thread-safe; calls on distinct contexts may run concurrently.
 *
 * There is NO CPU fallback: if no gfx950 device is usable fig_ctx_create() fails with
 * FIG_ENODEV.
 */
#ifndef FIGBIRD_HIP_H
#define FIGBIRD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIG_ABI_VERSION 1

#define FIG_OK 0
#define FIG_EINVAL (-1)   /* bad argument / inconsistent batch                      */
#define FIG_ENODEV (-2)   /* no usable HIP device (there is no CPU path)            */
#define FIG_ENOMEM (-3)   /* host or device allocation failed                       */
#define FIG_EHIP (-4)     /* a HIP runtime call or kernel failed                    */
#define FIG_ENOSPC (-5)   /* caller's result string buffer is too small             */
#define FIG_EUNSUP (-6)   /* input outside the supported envelope   */

typedef struct fig_ctx fig_ctx;

/* Model tables = the globals Figbird.cpp builds once per run from myout.sam / stat.txt
 * (A0: Figbird.cpp:47-91, built :846-921, :291-487, :497-844, :1156-1376, :7155-7200),
 * plus the run-level argv of Figbird.cpp main (:6957-6973).  Built on the host by the
 * caller (figfill does it in figbird_amd/csrc/host/fig_host.cpp: build_model). */
typedef struct fig_model {
    int32_t max_read_length;            /* maxReadLength (stat.txt col 3), <= 200            */
    const double *error_pos_dist;       /* errorPosDist[max_read_length]                    */
    const double *in_pos_dist;          /* inPosDist[max_read_length]                       */
    const double *del_pos_dist;         /* delPosDist[max_read_length]                      */
    double error_type_probs[25];        /* errorTypeProbs[5][5], row = from, col = to       */
    const double *insert_len_dist_smoothed; /* insertLengthDistSmoothed[max_insert_size]    */
    int32_t max_insert_size;            /* maxInsertSize                                    */
    int32_t insert_threshold_min;       /* insertThresholdMin (after the partial_len shift) */
    int32_t insert_threshold_max;       /* insertThresholdMax                               */
    int32_t gap_prob_cutoff;            /* gapProbCutOff                                    */
    /* run parameters */
    int32_t partial_flag;               /* argv[5]  */
    int32_t unmapped_flag;              /* argv[6]  */
    int32_t script_itr;                 /* argv[4]  */
    int32_t max_distance;               /* argv[2]  tempmaxDistance (D)                     */
    int32_t read_length;                /* argv[3]  unmapped_read_len                       */
    int32_t neg_overlap;                /* argv[12] gaplen_negative_overlap                 */
    int32_t partial_len;                /* argv[13]                                         */
    int32_t unm_limit;                  /* argv[14] gapthresh (400, FillGaps.cpp:22)        */
} fig_model;

/* One batch of gaps with their binned reads, as GapFiller sees them after
 * parseUnmapped/parsePartial (A2: Figbird.cpp:5661-5842).  SoA + CSR, host memory.
 * Sequences are ASCII; anything outside "ACGT" is treated as N. */
typedef struct fig_gap_batch {
    int64_t n_gaps;
    /* scaffold the gaps live in (Figbird.cpp:6979-7058): upper-case ASCII, concatenated */
    int64_t n_contigs;
    const int64_t *contig_off;          /* [n_contigs+1] offsets into contig_seq            */
    const char *contig_seq;
    /* per gap: gapInfo.txt (contigIdx, gapStart0, len) + stat2.txt + fillflag           */
    const int32_t *gap_contig;          /* [n_gaps]                                         */
    const int64_t *gap_start;           /* [n_gaps] 0-based start of the N run              */
    const int32_t *gap_len;             /* [n_gaps] originalGap                             */
    const int32_t *gap_stat2;           /* [n_gaps*3] fillNotfill, perfectReadGap, perfectReadGaplen */
    const int32_t *gap_fillflag;        /* [n_gaps] 1, or -1 when the gap had > 3000 read pairs (Figbird.cpp:7380-7385) */
    /* unmapped-mate reads (gaps_<g>.sam after parseUnmapped): CSR over gaps              */
    const int64_t *u_read_off;          /* [n_gaps+1] read index range per gap              */
    const int32_t *u_anchor_pos;        /* [n_ureads] pos_reads[] (1-based anchor position) */
    const uint8_t *u_is_reverse;        /* [n_ureads] isReverse[] (1 = mate was reverse-complemented) */
    const int64_t *u_seq_off;           /* [n_ureads+1] offsets into u_seq                  */
    const char *u_seq;                  /* reads_gap[] (already in gap orientation)         */
    /* partial (soft-clipped) reads (partial_gaps_<g>.sam): CSR over gaps, file order      */
    const int64_t *p_read_off;          /* [n_gaps+1]                                       */
    const int32_t *p_clipped_index;     /* [n_preads] col 2                                 */
    const int32_t *p_match;             /* [n_preads] col 3 (1/4 left side, 2/3 right side) */
    const int32_t *p_pos;               /* [n_preads] col 4                                 */
    const int32_t *p_ref_pos;           /* [n_preads] col 6 (mate position or -1)           */
    const int64_t *p_seq_off;           /* [n_preads+1] offsets into p_seq and p_qual       */
    const char *p_seq;
    const char *p_qual;                 /* phred+33, same offsets as p_seq (may be NULL in unmapped mode) */
    /* state the reference carries from gap to gap inside ONE worker process (FillGaps.cpp:456-649 deals the gaps to
     * $num_threads processes, each running Figbird.cpp main over its list in ascending gap order): [n_gaps] 1 when that
     * process has already set its global overlap_threshold (Figbird.cpp:103, :6317) by the time it gets to the gap.
     * NULL = the batch is one process taking the gaps in batch order (numthreads = 1): the library measures the carry
     * itself at upload.  A caller that deals the gaps to several processes, or fills a SHARD of a run (the gap that
     * set the value may be in another shard), uploads, asks fig_batch_probe_reach() which gaps get to :6317, carries
     * the bits along each process's gap list and hands the result to fig_batch_set_ot_preset() before the fill
     * (figfill: fig_host.cpp ot_presets_from_reach).  Only read in partial mode (the reference never reads the
     * global in an unmapped-mode run). */
    const uint8_t *gap_ot_preset;
} fig_gap_batch;

/* Results = what Figbird.cpp writes per gap: the gapout line (:7411-7413) and
 * gaptofill[g] (:6486-6492, :7463-7466). */
typedef struct fig_gap_results {
    int32_t *filled_len;                /* [n_gaps] gapStringLength                         */
    int32_t *gaptofill;                 /* [n_gaps] 0 or the negative-overlap length        */
    int64_t *str_off;                   /* [n_gaps+1] OUT: offsets of each gap string in str */
    char *str;                          /* OUT: concatenated gap strings (ACGTN), not NUL-terminated */
    int64_t str_capacity;               /* bytes available in str                           */
    /* optional parity/debug planes (NULL to disable): per gap, up to dbg_max_cand records
     * {gapEstimate, em_iterations, valid_count, likelihood} = likelihood_arr (Figbird.cpp:6390-6391) */
    int32_t dbg_max_cand;
    int32_t *dbg_n_cand;                /* [n_gaps]                                         */
    int32_t *dbg_cand_i;                /* [n_gaps*dbg_max_cand*3]                          */
    double *dbg_cand_lik;               /* [n_gaps*dbg_max_cand]                            */
    int32_t *dbg_n_place;               /* [n_gaps] placeReads invocations spent on the gap (needs dbg_n_cand)  */
    /* optional per-read final placements for draw.txt (Figbird.cpp:5140, 5381); NULL to disable */
    int32_t *draw_pos;                  /* [n_ureads + n_preads] maxPos-left_maxDistance, or INT32_MIN if the read was not drawn */
    int32_t *draw_isz;                  /* [n_ureads + n_preads] mleInsertSize / newInsertSize */
    int32_t *draw_len;                  /* [n_gaps*2] length argument of the unmapped / partial "S" header, -1 if absent */
    /* optional numeric planes of the parity contract (SURVEY.md §8 preamble), per gap and candidate INDEX j (candidate
     * length = gapMin + j; needs the dbg_cand buffers above; NULL to disable):
     *  (i)  countsGap[x][0..4] as the E-step of the candidate's LAST placeReads call left it (Figbird.cpp:3181-3187 /
     *       :3603-3611), x < min(candidate length, dbg_plane_cols);
     *  (ii) per read, the E-step's maximum log-likelihood over its placements, maxlikelihood_value[r] (:3258-3261 /
     *       :3680-3688; 0 when the read had no placement), r < min(reads of the gap, dbg_plane_reads). */
    int32_t dbg_plane_cols;
    int32_t dbg_plane_reads;
    double *dbg_counts;                 /* [n_gaps*dbg_max_cand*dbg_plane_cols*5]           */
    double *dbg_read_maxlv;             /* [n_gaps*dbg_max_cand*dbg_plane_reads]            */
} fig_gap_results;

/* Timing/occupancy facts of the last fig_fill_gaps call (for bench.py). */
typedef struct fig_stats {
    double kernel_ms;                   /* HIP-event time of the fill kernels on the library's stream */
    double h2d_ms, d2h_ms;              /* transfers (0 when the batch was already resident) */
    int64_t packed_bytes;               /* algorithmic input bytes uploaded (2-bit bases + masks + descriptors) */
    int64_t place_calls;                /* placeReads invocations executed on the device     */
    double alg_flops;                   /* algorithmic FP64 flops, counted on device */
    int32_t n_launches;
    int32_t pad0;
    double spec_flops;                  /* algorithmic flops of every candidate evaluation executed, discarded speculation included */
    double mle_alg_flops;               /* share of the executed evaluations' flops credited to the MLE passes (1 per placement and base) */
    double mle_exec_flops;              /* FP64 multiplies the MLE passes really executed (exact pruning skips the rest) */
} fig_stats;

int fig_version(void);
const char *fig_strerror(int code);

int fig_ctx_create(int device_ordinal, fig_ctx **out);
void fig_ctx_destroy(fig_ctx *ctx);
/* Sets the run-level model.  A batch left resident by fig_batch_upload is DROPPED (it was packed under the previous
 * model: launch classes, capacities and candidate ranges depend on it); upload again before fig_fill_resident. */
int fig_ctx_set_model(fig_ctx *ctx, const fig_model *model);

/* Worst-case bytes of results->str for this batch (sum of per-gap alloc_arg, Figbird.cpp:7395-7398). */
int64_t fig_results_capacity(const fig_model *model, const fig_gap_batch *batch);

/* Upload + pack a batch so that repeated fills run with inputs resident in HBM.
 * fig_fill_gaps(ctx, batch, out) == fig_batch_upload + fig_fill_resident + fig_batch_free. */
int fig_batch_upload(fig_ctx *ctx, const fig_gap_batch *batch);
int fig_fill_resident(fig_ctx *ctx, fig_gap_results *out);
void fig_batch_free(fig_ctx *ctx);
int fig_fill_gaps(fig_ctx *ctx, const fig_gap_batch *batch, fig_gap_results *out);

/* The exact carry of Figbird.cpp's process-global overlap_threshold (:103, :6298-6317; read at :2684, :2760-2766).
 * fig_batch_probe_reach: reach[g] = 1 iff the candidate loop of gap g of the RESIDENT batch gets to :6317 -- it does
 * unless the loop has no candidate, or its first initialize() leaves side_limit < 10 (:6303), or the gap closes by a
 * negative overlap at its first candidate (:6305-6306); measured on the device before any gap is filled (none of it
 * depends on the threshold).  Unmapped-mode runs never read the global: reach is all 0 there and nothing is launched.
 * fig_batch_set_ot_preset: replaces gap_ot_preset of the resident batch ([n_gaps], batch order). */
int fig_batch_probe_reach(fig_ctx *ctx, uint8_t *reach);
int fig_batch_set_ot_preset(fig_ctx *ctx, const uint8_t *preset);

int fig_get_stats(const fig_ctx *ctx, fig_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* FIGBIRD_HIP_H */
