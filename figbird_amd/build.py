"""Build every native artefact in-tree (no JIT cache): libfighip.so (gfx950 HIP engine + C ABI),
figfill (C++ host drop-in for FillGaps.cpp), and -- test infrastructure only -- the oracle
restatement, the reference binaries (when /root/reference is present) and the one-lane
emulation build of figfill used by the CPU unit tests."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
BINDIR = os.path.join(PKG, "bin")
LIB = os.path.join(LIBDIR, "libfighip.so")
FIGFILL = os.path.join(BINDIR, "figfill")
HOSTLIB = os.path.join(LIBDIR, "libfighost.so")
EMU = os.path.join(ROOT, "tests", "emu", "figfill_emu")
EMULIB = os.path.join(ROOT, "tests", "emu", "libfigemu.so")   # test-only: C ABI backed by the one-lane emulation
ORACLE = os.path.join(ROOT, "oracle", "figbird_oracle")
REFDIR = os.path.join(ROOT, "oracle", "_ref")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the reference multiplies and adds separately (no FMA on its x86-64 build);
# bit-identical pile-up weights need the same two roundings.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-value"]


def _csrc_files():
    """Every source the device library / emulation depends on (all of csrc/ + the ABI header)."""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".cpp"))]
    out.append(os.path.join(ROOT, "include", "figbird_hip.h"))
    return out


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]}")
    return r


def build_lib(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _csrc_files()
    if force or not _newer(LIB, srcs):
        _run([HIPCC] + HIP_FLAGS + ["-o", LIB, os.path.join(CSRC, "fig_abi.hip")])
    return LIB


def build_figfill(force: bool = False) -> str:
    os.makedirs(BINDIR, exist_ok=True)
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(host, f) for f in ("figfill_main.cpp", "fig_host.cpp", "fig_host.h")]
    if force or not _newer(FIGFILL, srcs + [LIB]):
        _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", FIGFILL,
              os.path.join(host, "figfill_main.cpp"), os.path.join(host, "fig_host.cpp"),
              "-L" + LIBDIR, "-lfighip", "-Wl,-rpath,$ORIGIN/../lib"])
    if force or not _newer(HOSTLIB, srcs):
        _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", HOSTLIB, os.path.join(host, "fig_host.cpp")])
    return FIGFILL


def build_test_infra(force: bool = False) -> None:
    """oracle/ restatement, oracle/_ref (only where /root/reference exists) and the emu figfill."""
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    if os.path.exists("/root/reference/Figbird.cpp"):
        need = force or not all(os.path.exists(os.path.join(REFDIR, f)) for f in ("Figbird.out", "Figbird_O0.out", "FillGaps.out"))
        if need:
            _run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(host, "figfill_main.cpp"), os.path.join(host, "fig_host.cpp"), os.path.join(host, "fig_host.h"),
            os.path.join(ROOT, "tests", "emu", "fig_emu_abi.cpp")] + _csrc_files()
    if force or not _newer(EMU, srcs):
        _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", EMU, srcs[0], srcs[1], srcs[3]])
    if force or not _newer(EMULIB, srcs):
        _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", EMULIB, srcs[3]])


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_figfill(force)
    build_test_infra(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built:", LIB, FIGFILL, ORACLE, EMU)
