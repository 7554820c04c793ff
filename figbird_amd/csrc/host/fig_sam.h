// fig_sam.h -- SAM ingest + read -> gap binning in the C++ host: the counterpart of Preprocess.cpp
// (main :1832-2676; printVectors :641-855, printMixedVectors :999-1489, checkPos :536-614, checkPos2 :616-639,
// writePartialSam :425-502, collectPartialSAM :1667-1694, checkMIM :885-925, check_duplicate :362-402).
// Same 13 arguments, same files: Temp/{gapInfo,stat,stat2}.txt, Alignments/myout.sam, Gaps/{gaps_N,partial_gaps_N}.sam
// (+ the *_reduced read files when the driver asks for them).  Reads are binned through a per-contig interval index over
// the gaps instead of the reference's scan of every gap per read, and the per-gap files are written once at the end
// instead of one fopen/fclose per read; the bytes are the same.
#ifndef FIG_SAM_H
#define FIG_SAM_H
#include <cstdint>
#include <string>
#include <vector>

namespace figsam {

struct Args {                       // Preprocess.cpp:1862-1874 (argv[1..13])
    std::string contigFile; int maxDistance = 0; int samflag = 0; std::string mapFile, outFile, filledContigFile, reads1, reads2, gapsDir, tmpDir;
    int default_setting = 0, genome_reduction = 0, read_reduction = 0;
};

struct GapRec { int contigNo = 0; long gapStart = 0; int gapLength = 0; int contigToWrite = 0; };

// What the binning produced, kept in memory as well as written to files: the CSR a caller can hand straight to the fill.
struct Binned {
    std::vector<GapRec> gaps;
    std::vector<std::string> gap_files;        // text of gaps_<g>.sam          (samflag 2)
    std::vector<std::string> partial_files;    // text of partial_gaps_<g>.sam  (samflag 1)
    std::vector<int> perfect_gap, perfect_len; // stat2.txt columns 2, 3 (checkMIM)
    long totalCount = 0, unCount = 0; unsigned long maxReadLength = 0;
    std::vector<std::string> stdout_lines;     // what the reference prints (the driver captures it: RunFigbird.sh:285)
};

// Runs the whole stage; returns the process exit code (0 ok, 1 on the reference's "Can't open ..." errors).
// write_mode: 1 = every file the reference writes; 2 = only the run-level files (gapInfo/stat/stat2/myout + reduced reads),
// the per-gap texts staying in `out` for an in-memory hand-over to the fill; 0 = nothing.
int preprocess(const Args &a, Binned &out, std::string &err, int write_mode = 1);
// argv[0] = "preprocess", argv[1..13] = the reference's arguments
int preprocess_main(int argc, char **argv);

}  // namespace figsam
#endif
