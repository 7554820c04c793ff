// figtool -- the small stages RunFigbird.sh compiles and runs around the fill, as ONE prebuilt host binary:
//   figtool flanktrim  <genome> <trim> <maxreadlen> <out>        replaces `g++ FlankTrim.cpp && ./a.out ...`   (RunFigbird.sh:254,433)
//   figtool rewrap     <in> <out> <columns>                      replaces `python reference.py ...`            (:256,435,809)
//   figtool reduce-scf <genome> <tmp/>                           replaces `g++ Reduce_SCF.cpp && ./a.out ...`  (:266,320)
//   figtool combine    <num_itr> <Individual_gaps/>              replaces `g++ CombineGaps.cpp && ./a.out ...` (:777)
//   figtool preprocess <13 Preprocess.cpp arguments>             replaces `g++ Preprocess.cpp && ./a.out ...`  (:285,338,451,472)
// Same arguments, same files, same bytes; errors go to stderr with exit code 1 as the reference's do.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "fig_plumb.h"
#include "fig_sam.h"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: figtool flanktrim|rewrap|reduce-scf|combine|preprocess <args...>\n"); return 1; }
    const std::string cmd = argv[1];
    std::string err;
    if (cmd == "flanktrim" && argc >= 6) {
        if (!figplumb::flank_trim(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5], err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        return 0;
    }
    if (cmd == "rewrap" && argc >= 5) {
        if (!figplumb::rewrap_fasta(argv[2], argv[3], atoi(argv[4]), err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        return 0;
    }
    if (cmd == "reduce-scf" && argc >= 4) {
        if (!figplumb::reduce_scf(argv[2], argv[3], err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        return 0;
    }
    if (cmd == "combine" && argc >= 4) {
        int rc = figplumb::combine_gaps(atoi(argv[2]), argv[3], err);
        if (rc) printf("%s\n", err.c_str());
        return rc;
    }
    if (cmd == "preprocess" && argc >= 15) return figsam::preprocess_main(argc - 1, argv + 1);
    fprintf(stderr, "figtool: unknown command or too few arguments: %s\n", cmd.c_str());
    return 1;
}
