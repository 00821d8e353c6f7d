"""Build libmi355det.so (HIP, gfx950 only) in-tree with hipcc.

    python -m object_detectors_amd.build            # incremental
    python -m object_detectors_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmi355det.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# (source, extra flags).  Box/criterion kernels need unfused IEEE float32 for bit-exact index
# decisions; the MFMA files are free to contract.
SOURCES = [
    ("lib.cpp", []),
    ("yolo_kernels.hip", ["-ffp-contract=off"]),
    ("box_kernels.hip", ["-ffp-contract=off"]),
    ("roi_kernels.hip", ["-ffp-contract=off"]),
    ("proposal_kernels.hip", ["-ffp-contract=off"]),
    ("frcnn_kernels.hip", ["-ffp-contract=off"]),
    ("transform_kernels.hip", ["-ffp-contract=off"]),
    ("resnet_kernels.hip", []),
    ("conv_kernels.hip", []),
    ("igemm8_kernels.hip", []),
    ("wgrad_kernels.hip", []),
    ("dgrad_s2_kernels.hip", []),
    ("elem_kernels.hip", []),
    ("stem_kernels.hip", ["-fno-slp-vectorize"]),
    ("stem_l1_kernels.hip", ["-fno-slp-vectorize"]),     # packed fp32 adds cost more moves than they save (and 36 VGPRs)
    ("rstem_kernels.hip", ["-fno-slp-vectorize"]),
]
# Files that touch stored activations are compiled a second time with -DMI355_F16=1 (fp16 storage instead of bf16: csrc/common.h); the
# entry points of those objects carry the suffix _f16 (csrc/f16_names.h, include/mi355det_f16.h).
F16_TWINS = ["conv_kernels.hip", "igemm8_kernels.hip", "wgrad_kernels.hip", "dgrad_s2_kernels.hip", "elem_kernels.hip", "stem_kernels.hip",
             "stem_l1_kernels.hip"]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value", "-x", "hip"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "mi355det.h"))
    jobs, objs = [], []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError(sp)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            jobs.append([HIPCC] + COMMON + extra + ["-c", sp, "-o", op])
        if src in F16_TWINS:
            op16 = os.path.join(OBJ, os.path.splitext(src)[0] + "_f16.o")
            objs.append(op16)
            if force or _stale(op16, [sp] + headers):
                jobs.append([HIPCC] + COMMON + extra + ["-DMI355_F16=1", "-c", sp, "-o", op16])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
