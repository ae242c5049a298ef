"""Build libd3ga_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["deform.hip", "raster_pre.hip", "raster_bin.hip", "raster_composite.hip", "raster_composite_scan.hip", "raster_api.hip", "bary.hip", "loss.hip", "mlp.hip", "encoding.hip"]
LINK_MAP = os.path.join(HERE, "d3ga.map")
HEADERS = ["d3ga_math.h", "d3ga_internal.h", "raster_pre_body.h", "composite_common.h", os.path.join("..", "..", "include", "d3ga.h")]
ABL = os.environ.get("D3GA_SCAN_ABL")       # timing ablation of the compositing backward (wrong results): own objects + .so
VARIANT = os.environ.get("D3GA_VARIANT")    # A/B build of compile-time knobs: "tag:-DNAME=value,-DOTHER=value" -> tools/_build/libd3ga_hip_<tag>.so (correct results)
DIAG = os.environ.get("D3GA_DIAG") or (("abl" + ABL) if ABL else None) or (("var" + VARIANT.split(":")[0]) if VARIANT else None)          # diagnostic build: its own objects and its own .so (D3GA_LIB_PATH selects it)
# Diagnostic / ablation builds never land in the package directory: tools/_build/ (git-ignored, travels with gpurun).
# _lib.py refuses an ablation build (d3ga_debug_defaults()[0] != 0: WRONG results by design) unless D3GA_ALLOW_ABLATION=1.
DIAG_DIR = os.path.abspath(os.path.join(HERE, "..", "..", "tools", "_build"))
_DIAG_TAG = (VARIANT.split(":")[0] if (VARIANT and not os.environ.get("D3GA_DIAG") and not ABL) else
             "diag" if os.environ.get("D3GA_DIAG") in (None, "counters") else os.environ["D3GA_DIAG"])
if VARIANT:
    FLAGS_VARIANT = [f for f in VARIANT.split(":", 1)[1].split(",") if f]
else:
    FLAGS_VARIANT = []
OUT = (os.path.join(DIAG_DIR, f"libd3ga_hip_abl{ABL}.so" if ABL else f"libd3ga_hip_{_DIAG_TAG}.so") if DIAG
       else os.path.join(HERE, "..", "libd3ga_hip.so"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"] + FLAGS_VARIANT
# per-source extras.  The entry-per-lane compositing backward is VALU-issue bound; SLP-packing its scalar f32 chains into
# v_pk_* costs ~35 register shuffles per 4 pixels (ISA inspected) and a v_pk_fma_f32 issues in 4.2 cycles against 2.4 for a
# v_fma_f32 (tools/micro/valu_issue.hip), so the vectoriser is off for the two compositing files.
EXTRA = {"raster_composite_scan.hip": ["-fno-slp-vectorize"],
         "raster_composite.hip": ["-fno-slp-vectorize"]}      # forward 117 -> 103 us at C3: packed f32 ops cost 2x, plus their shuffles
if os.environ.get("D3GA_CHAIN_WAVES"):                        # A/B: wavefronts per workgroup of the fused field-network kernel
    FLAGS.append("-DD3GA_CHAIN_WAVES=" + os.environ["D3GA_CHAIN_WAVES"])
if os.environ.get("D3GA_CHAIN_CS"):                           # A/B: k-steps per weight chunk of the fused field-network kernel
    FLAGS.append("-DD3GA_CHAIN_CS=" + os.environ["D3GA_CHAIN_CS"])
if os.environ.get("D3GA_ALL_NOSLP"):                          # A/B: every translation unit
    FLAGS.append("-fno-slp-vectorize")
if ABL:
    FLAGS.append("-DD3GA_SCAN_ABL=" + ABL.split("w")[0].split("p")[0].split("f")[0])
    if "w" in ABL:                         # e.g. D3GA_SCAN_ABL=0w5: no ablation, register budget for 5 wavefronts per SIMD
        FLAGS.append("-DD3GA_TILE_WAVES=" + ABL.split("w")[1])
    if "f" in ABL:                         # e.g. D3GA_SCAN_ABL=0f6: two-stage forward with the register budget of 6 wavefronts per SIMD
        FLAGS.append("-DD3GA_FWD_WAVES=" + ABL.split("f")[1])
    if "p" in ABL:                         # e.g. D3GA_SCAN_ABL=0p: s_setprio by remaining groups
        FLAGS.append("-DD3GA_TILE_PRIO=1")
if os.environ.get("D3GA_DIAG"):            # diagnostic build (loop statistics and per-wave timeline, tools/diag_scan.py); never the shipped one
    FLAGS += ["-DD3GA_DIAG", "-DD3GA_DIAG_TIMELINE"]       # D3GA_DIAG=timeline: per-wave start / end / trip counts only (few registers)
    if os.environ.get("D3GA_DIAG") == "counters":          # + loop statistics (lane efficiency, cache hits, ...): perturbs the timing
        FLAGS.append("-DD3GA_DIAG_COUNTERS")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(DIAG_DIR, f"obj_abl{ABL}" if ABL else f"obj_{_DIAG_TAG}") if DIAG else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _newer(OUT, objs + [LINK_MAP]):
        # Link against the HIP runtime that PyTorch-ROCm itself loads (torch/lib/libamdhip64.so, SONAME without a
        # version) so that the process holds ONE runtime: torch's streams, events and allocations are then valid
        # in our launches.  /opt/rocm/lib stays on the runpath for hosts that load the library without torch.
        import torch
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        cmd = ["g++", "-shared", "-fPIC"] + objs + ["-L" + tlib, "-lamdhip64", "-Wl,-rpath," + tlib,
                                                    "-Wl,-rpath,/opt/rocm/lib", "-Wl,--version-script=" + LINK_MAP, "-o", OUT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return os.path.abspath(OUT)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
