// fig_gaprules.h -- per-gap rules of the reference that both the packer (fig_pack.h, inside libfighip.so) and the C++
// host (figbird_amd/csrc/host) need: the candidate range of a gap and the carry rule of the one piece of state the
// reference hands from gap to gap inside a worker process.  Pure C++, no device or host-library dependencies.
#ifndef FIG_GAPRULES_H
#define FIG_GAPRULES_H
#include <algorithm>
#include <stdint.h>
#include "../../include/figbird_hip.h"

// findFrac + alloc_arg, Figbird.cpp:6879-6906, :7393-7400 (float arithmetic as in the reference)
static void gap_alloc(const fig_model *m, int G0, int *alloc_arg, float *f1, float *f2, int *lgf) {
    int factor = 3 * m->partial_len;
    int mid_limitp = 2 * m->partial_len, mid_limitu = m->unm_limit;
    float info[3] = {0, 0, 0};
    int ret;
    if (m->partial_flag) {
        if (G0 <= mid_limitp / 2) { info[0] = .00001; info[1] = (float)factor / G0; ret = -1; }
        else if (G0 <= mid_limitp) { info[0] = .00001; info[1] = 5.0; ret = 5; }
        else { info[0] = 1; info[1] = 1; ret = 3; }
    } else {
        if (G0 <= mid_limitu / 3) { info[0] = .3; info[1] = (float)factor / G0; ret = -1; }
        else if (G0 <= mid_limitu) { info[0] = .5; info[1] = 2.5; ret = 3; }
        else { info[0] = 1; info[1] = 1; info[2] = 1; ret = 1; }
    }
    *alloc_arg = (ret == -1) ? factor * 3 : G0 * ret;
    *f1 = info[0]; *f2 = info[1]; *lgf = (int)info[2];
}

static int gap_range(int G0, float f1, float f2) {
    int gapMin = (int)(G0 * f1), gapMax = (int)(G0 * f2);
    int r = gapMax - gapMin + 1;
    return r < 1 ? 1 : r;
}

// The reference's `overlap_threshold` is a file-level global of Figbird.cpp (:103), 0 when a worker process starts and set
// to 5 the first time a candidate loop of that process gets to :6317; every later gap OF THAT PROCESS sees 5 from its first
// line on.  Whether a gap's loop gets there is measured on the device (`reach`, fig_engine_sched.h: fig_gap_probe -- it
// depends on the gap's reads: a gap that closes by a negative overlap at its first candidate leaves the loop before that
// line); the carry is the prefix-OR of those bits along the process's gap list (ascending gap ids, Figbird.cpp:7277-7317).
// `process`: gap ids in the order the process takes them; reach/preset are indexed by gap id.
template <class IdList>
static inline void fig_ot_carry(const IdList &process, const uint8_t *reach, uint8_t *preset) {
    bool reached = false;
    for (auto g : process) { preset[(size_t)g] = reached ? 1 : 0; if (reach[(size_t)g]) reached = true; }
}

#endif
