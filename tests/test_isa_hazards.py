"""CPU: the built library's device code holds no packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 are fp32-pair
ops of gfx950).  Round 5 found that on the MI355X a packed-fp32 instruction with non-default operand selects -- what the compiler makes of
"pair (op) broadcast scalar" -- computes lanes 48-63 with the default selects when a wave of ANOTHER kernel on the same SIMD issues an f16 matrix
instruction beside it (tools/pk_mfma_hazard.hip reproduces it with no product code; HISTORY.md section 4).  The forward overlaps its side-stream
kernels (FPS, the kNN head, the GMM E/M, the clustering loss) with fp16x3 GEMMs, so the library is built without the packed forms
(ogmm_amd/csrc/Makefile) and this test keeps it that way."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from ogmm_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def _device_disassembly(lib_path):
    """Disassembly of every gfx950 code object bundled in the shared library (llvm-objdump --offloading writes the bundles beside its input, so the
    library is copied to a scratch directory first)."""
    with tempfile.TemporaryDirectory() as tmp:
        work = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, work)
        subprocess.run([OBJDUMP, "--offloading", work], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f)
        assert objs, "no gfx950 code object found in %s" % lib_path
        text = []
        for f in objs:
            text.append(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(text), len(objs)


@pytest.fixture(scope="module")
def disassembly():
    if not (os.path.isfile(OBJDUMP) and os.access(OBJDUMP, os.X_OK)):
        pytest.skip("llvm-objdump not in this image")
    import __graft_entry__ as g
    if not os.path.isfile(_lib.LIB_PATH):
        g.build()
    return _device_disassembly(_lib.LIB_PATH)


def test_library_holds_device_code_for_every_source(disassembly):
    text, n_objs = disassembly
    kernels = set(re.findall(r"^[0-9a-f]+ <(_Z\w+)>:", text, re.M))
    assert n_objs >= 15 and len(kernels) >= 100, (n_objs, len(kernels))
    assert any("fps_kernel" in k for k in kernels) and any("gemm_f16x3_v10_kernel" in k for k in kernels)
    assert "v_mfma_f32_32x32x16_f16" in text          # (the scan below is not vacuous: this is gfx950 code with its matrix instructions)


def test_no_packed_fp32_instruction_in_any_kernel(disassembly):
    text, _ = disassembly
    cur, hits = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
            hits.append((cur, line.split("//")[0].strip()))
    assert not hits, "packed-fp32 instructions in the device code (see ogmm_amd/csrc/Makefile): %s" % hits[:8]


def test_operand_selects_only_on_the_instructions_checked_on_the_part(disassembly):
    """The hazard lives in the operand-select bits of the VOP3P encoding.  Of the instructions that carry them, tools/pk_mfma_hazard.hip has checked
    v_fma_mix_f32 and the v_fma_mixlo_f16 / v_fma_mixhi_f16 pair (the engines' in-register binary16 split) beside f16 matrix neighbours: not affected
    (profiles/round5_pk_mfma_hazard*.txt).  Anything else with a select modifier -- packed f16 / i16 arithmetic, dot products -- has NOT been checked
    and must not appear in the library without such a check."""
    text, _ = disassembly
    checked = ("v_fma_mix_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16")
    other = set()
    for line in text.splitlines():
        if "op_sel" in line:
            op = line.split("//")[0].split()
            op = next((w for w in op if w.startswith(("v_", "s_", "ds_", "global_", "buffer_"))), "?")
            if op not in checked:
                other.add(op)
    assert not other, "instructions with operand-select modifiers that were never checked beside f16 matrix instructions: %s" % sorted(other)
