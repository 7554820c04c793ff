// fig_host.h -- host side of the drop-in `figfill` (replaces FillGaps.cpp + Figbird.cpp main):
// file ingest at the post-Preprocess boundary, the run-level error /
// insert-size model (A0), and the output writers (gapout.txt, draw.txt, filledContigs.fa,
// Ncount.txt).  No arithmetic of the per-gap hot path lives here: that is libfighip.so.
#ifndef FIG_HOST_H
#define FIG_HOST_H
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include "../../../include/figbird_hip.h"

namespace fighost {

struct RunArgs {                    // FillGaps.cpp:419-433
    std::string contigFile; int D = 0; int read_length = 0; int script_itr = 0; int partial_flag = 0;
    int unmapped = 0; int num_threads = 1; std::string mapFile, tmp, gapsDir; int neg_overlap = 0;
    int partial_len = 0; int trim = 0; int setinputmean = 0; int isz = 0; int unm_limit = 400;
};

struct Scaffold {
    std::vector<std::string> names;
    std::vector<int64_t> off;       // [n+1]
    std::string seq;                // concatenated, upper-case
    int64_t n() const { return (int64_t)names.size(); }
};

// Figbird.cpp:6979-7058
bool load_scaffold(const std::string &path, Scaffold &sc, std::string &err);

struct Model {                      // owns the tables fig_model points into
    std::vector<double> errorPosDist, inPosDist, delPosDist, insertLengthDistSmoothed;
    double errorTypeProbs[25];
    int maxReadLength = 0, maxInsertSize = 0, Tmin = 0, Tmax = 0, cutoff = 0;
    double insertSizeMean = 0, leftSD = 0, rightSD = 0;
    void fill(fig_model &m, const RunArgs &a) const;
};

// A0: Figbird.cpp:7084-7200 (processMapping, computeProbabilites, computeLikelihood, cutoff, thresholds)
bool build_model(const RunArgs &a, const Scaffold &sc, Model &out, std::string &err);

struct Batch {                      // owns the arrays fig_gap_batch points into
    std::vector<int32_t> gap_contig, gap_len, gap_stat2, gap_fillflag;
    std::vector<int64_t> gap_start;
    std::vector<int64_t> u_read_off, u_seq_off, p_read_off, p_seq_off;
    std::vector<int32_t> u_anchor_pos, p_clipped_index, p_match, p_pos, p_ref_pos;
    std::vector<uint8_t> u_is_reverse;
    std::string u_seq, p_seq, p_qual;
    std::vector<std::string> messages;      // stdout lines the reference prints while parsing
    // the reference's worker processes (assign_processes): gap ids per process in the order it fills them, and per gap
    // whether its process has set overlap_threshold before it gets there (fig_gap_batch::gap_ot_preset; measured, see below)
    std::vector<std::vector<int32_t>> processes;
    std::vector<uint8_t> gap_ot_preset;
    void view(fig_gap_batch &b, const Scaffold &sc) const;
};

// FillGaps.cpp:456-649 + writeGapLoad :313-334: the gaps each of the $num_threads worker processes fills, ascending within
// a process.  (Gaps of at most `gapthresh` = 400 bp are dealt round-robin, the longer ones fill the processes up in the
// order of their remaining capacity.)
std::vector<std::vector<int32_t>> thread_allocation(const std::vector<int32_t> &gap_len, int num_threads, int gapthresh = 400);
// Fills B.processes for the run `a` (needs only the run parameters, not the model tables); B.gap_ot_preset is all 0 until
// ot_presets_from_reach() has the measured bits (fig_batch_probe_reach, include/figbird_hip.h), indexed by gap id.
void assign_processes(const RunArgs &a, const Scaffold &sc, Batch &B);
void ot_presets_from_reach(Batch &B, const uint8_t *reach);
// gaploads.txt as the reference leaves it in Temp/ (one line of tab-terminated gap ids per process)
bool write_gaploads(const RunArgs &a, const Batch &b, std::string &err);

// gapInfo.txt, stat2.txt, gaps_<g>.sam (parseUnmapped, Figbird.cpp:5661-5767), partial_gaps_<g>.sam
bool load_batch(const RunArgs &a, const Scaffold &sc, Batch &out, std::string &err);

// The same parser fed from memory instead of per-gap files: `gaps_text[g]` / `partial_text[g]` are the texts the binning
// stage (fig_sam.h) would have written to gaps_<g>.sam / partial_gaps_<g>.sam (either may be empty = "no such files":
// unmapped-mode reads are then absent, partial files count as empty).  gapInfo.txt / stat2.txt still come from a.tmp.
bool load_batch_mem(const RunArgs &a, const Scaffold &sc, const std::vector<std::string> *gaps_text, const std::vector<std::string> *partial_text,
                    Batch &out, std::string &err);

// ---- N GPUs from the C++ host (figfill with FIGFILL_DEVICES=0,1,...): the role of FillGaps.cpp:456-649 (deal the gaps) and
// :668-679 (start the workers).  Estimated cost of every gap (R * W * L * candidates * placeReads calls, the candidate range of
// findFrac, Figbird.cpp:6879-6906; same formula as figbird_amd/dist.py), a longest-processing-time-first deal into `world`
// shards (ascending gap ids inside a shard), and a shard of a batch as a batch of its own (gap_ot_preset travels with it).
std::vector<double> estimate_cost(const Batch &b, const RunArgs &a, int max_read_length);
std::vector<std::vector<int64_t>> partition_lpt(const std::vector<double> &cost, int world);
void make_shard(const Batch &b, const std::vector<int64_t> &ids, Batch &out);

struct Results {
    std::vector<int32_t> filled_len, gaptofill, draw_pos, draw_isz, draw_len;
    std::vector<int64_t> str_off;
    std::string str;
};

// gapout.txt (Figbird.cpp:7413 + FillGaps.cpp:140-219), draw.txt (draw_read, Figbird.cpp:2385-2427)
bool write_gapout(const RunArgs &a, const Batch &b, const Results &r, std::string &err);
bool write_draw(const RunArgs &a, const Batch &b, const Results &r, std::string &err);
// filledContigs.fa + Ncount.txt (FillGaps.cpp:708-926)
bool write_scaffold(const RunArgs &a, const Scaffold &sc, const Batch &b, const Results &r, std::string &err);

}  // namespace fighost
#endif
