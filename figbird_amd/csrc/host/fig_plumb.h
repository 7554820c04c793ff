// fig_plumb.h -- scaffold / report plumbing of the Figbird pipeline restated in the C++ host.
#ifndef FIG_PLUMB_H
#define FIG_PLUMB_H
#include <string>

namespace figplumb {
// FlankTrim.cpp:21-233: N-mask `trimsize` clean bases either side of every gap of 2..readlen-1 N; one line per contig.
bool flank_trim(const std::string &in, int trimsize, int readlen, const std::string &out, std::string &err);
// reference.py:1-29: re-wrap each sequence line to `slice` columns (with its blank-line and no-trailing-newline quirks).
bool rewrap_fasta(const std::string &in, const std::string &out, int slice, std::string &err);
// Reduce_SCF.cpp:14-152: keep only records that contain N -> <tmp_dir>newgenome.fa.
bool reduce_scf(const std::string &in, const std::string &tmp_dir, std::string &err);
// CombineGaps.cpp:169-313: stitch gapout_1..k.txt into combined_gapstring.txt / Individual_gaps.txt; returns the exit code.
int combine_gaps(int num_itr, const std::string &path, std::string &err);
}  // namespace figplumb
#endif
