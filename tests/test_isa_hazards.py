"""Build-time check of a gfx950 hazard the round-3 stem kernels ran into: a 16-byte `buffer_store_dwordx4` whose tile offset
sits in the SGPR `soffset` got no wait state before the next instruction rewrote its data VGPRs - 4 wrong pixels per tile, only when another process
shared the GPU.  The kernels carry the tile offset in the VECTOR offset instead (soffset 0), for which LLVM's hazard recogniser inserts the wait; nothing
in the source pins that, so this test disassembles the device code of every file that issues buffer stores and asserts, for each wide store,
  * soffset is the literal 0 (a compiler change that folds a uniform offset back into an SGPR fails here, not on a shared GPU), and
  * the instruction right behind the store does not write one of its data registers (the wait state is there).
hipcc cross-compiles for gfx950 without a GPU (1-2 s per file with --cuda-device-only -S)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "object_detectors_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILES = [("stem_kernels.hip", ["-fno-slp-vectorize"]), ("stem_l1_kernels.hip", ["-fno-slp-vectorize"])]

STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(v\d+|off),\s*s\[\d+:\d+\],\s*(\S+)")
DEST = re.compile(r"^\s*(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*|v_mfma\w*)\s+(v(\d+)|v\[(\d+):(\d+)\])")


def _asm(src, extra, tmp_path, defs=()):
    out = os.path.join(str(tmp_path), os.path.splitext(src)[0] + ("_f16" if defs else "") + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-x", "hip", "--cuda-device-only", "-S"] + extra + list(defs) + \
          [os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines()


@pytest.mark.parametrize("src,extra", FILES)
@pytest.mark.parametrize("defs", [(), ("-DMI355_F16=1",)], ids=["bf16", "fp16"])
def test_wide_buffer_stores_have_no_scalar_offset_and_a_wait_state(src, extra, defs, tmp_path):
    lines = [l for l in _asm(src, extra, tmp_path, defs) if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.rstrip().endswith(":")]
    stores = 0
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m:
            continue
        stores += 1
        lo, hi, soff = int(m.group(2)), int(m.group(3)), m.group(5).rstrip(",")
        assert soff == "0", f"{src}: wide buffer store with a scalar offset register ({soff}): {l.strip()}"
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        d = DEST.match(nxt)
        if d:
            a, b = (int(d.group(3)), int(d.group(3))) if d.group(3) else (int(d.group(4)), int(d.group(5)))
            assert b < lo or a > hi, f"{src}: the instruction behind a wide buffer store rewrites its data registers without a wait state:\n{l}\n{nxt}"
    assert stores >= 3, f"{src}: expected the 16-byte buffer stores of the stem kernels in the disassembly, found {stores}"
