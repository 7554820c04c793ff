"""Build the TEST INFRASTRUCTURE (never shipped, never imported by the product package figbird_amd/):
the oracle restatement (oracle/figbird_oracle), the reference binaries compiled from the sources where they lie
(oracle/_ref/, only where /root/reference exists -- the GPU box uses the prebuilt files), and the one-lane host
emulation of the device engine used by the CPU unit tests (tests/emu/)."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from figbird_amd import build as fbuild  # noqa: E402

EMU = os.path.join(ROOT, "tests", "emu", "figfill_emu")
EMULIB = os.path.join(ROOT, "tests", "emu", "libfigemu.so")   # test-only: C ABI backed by the one-lane emulation
ORACLE = os.path.join(ROOT, "oracle", "figbird_oracle")
REFDIR = os.path.join(ROOT, "oracle", "_ref")
REF_BINARIES = ("Figbird.out", "Figbird_O0.out", "FillGaps.out", "Preprocess.out", "CombineGaps.out", "FlankTrim.out", "Reduce_SCF.out")


def build(force: bool = False) -> None:
    fbuild._run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    if os.path.exists("/root/reference/Figbird.cpp"):
        need = force or not all(os.path.exists(os.path.join(REFDIR, f)) for f in REF_BINARIES)
        if need:
            fbuild._run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    host = os.path.join(fbuild.CSRC, "host")
    hsrcs = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith((".cpp", ".h")))
    main_cpp = os.path.join(host, "figfill_main.cpp")
    host_cpps = [f for f in hsrcs if f.endswith(".cpp") and os.path.basename(f).startswith("fig_")]
    emu_abi = os.path.join(ROOT, "tests", "emu", "fig_emu_abi.cpp")
    srcs = hsrcs + [emu_abi] + fbuild._csrc_files()
    if force or not fbuild._newer(EMU, srcs):
        fbuild._run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", EMU, main_cpp] + host_cpps + [emu_abi])
    if force or not fbuild._newer(EMULIB, srcs):
        fbuild._run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", EMULIB, emu_abi])


if __name__ == "__main__":
    build("--force" in sys.argv)
    print("built:", ORACLE, EMU, EMULIB)
