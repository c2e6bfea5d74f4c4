"""Build-time properties of the gfx950 code object, checked on the CPU (hipcc cross-compiles without a GPU): no kernel may spill
to scratch memory (round 3: a not-inlined lambda once put a kernel's whole state there -- parity tests still pass, at 1/8 of the
speed), the fill kernels must use the instructions DESIGN.md says they use, and the headline kernel must keep the register count
that gives it four waves per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dpx_gpu_genomics_project_amd", "csrc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    out = tmp_path_factory.mktemp("isa") / "dpx_kernels.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                    os.path.join(CSRC, "dpx_kernels.hip"), "-o", str(out)], check=True, timeout=600)
    return open(out).read()


def _kernels(isa):
    """{mangled name: (metadata text, body text)}"""
    meta = {m.group(1): m.group(0) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?.*\.vgpr_count:\s+\d+", isa)}
    out = {}
    for name in meta:
        start = isa.find("\n" + name + ":")
        end = isa.find(".Lfunc_end", start)
        out[name] = (meta[name], isa[start:end] if start >= 0 else "")
    return out


def test_no_kernel_uses_scratch_memory(isa):
    ks = _kernels(isa)
    assert len(ks) > 60, len(ks)
    for name, (meta, body) in ks.items():
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", meta), name
        assert "scratch_" not in body, name


def test_fill_kernels_use_the_instructions_the_design_names(isa):
    ks = _kernels(isa)
    pick = lambda frag: next(v for k, v in ks.items() if frag in k)
    # headline: packed LSW, 16 rows per lane, row-tag keys -- VOP3P cell update, DPP hand-over, 16-byte stores, <= 104 VGPRs (4 waves / SIMD)
    meta, body = pick("k_linear_fill_pkILi16ELb1ELb1EE")
    for ins in ("v_pk_max_i16", "v_pk_mad_i16", "v_pk_add_u16", "v_pk_min_u16", "v_pk_mad_u16", "v_pk_max_u16", "v_mov_b32_dpp", "global_store_dwordx4"):
        assert ins in body, ins
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 104
    # the per-row-key variant (scores too large for the tags) is the round-2 kernel: more registers, still no spills
    meta, _ = pick("k_linear_fill_pkILi16ELb1ELb0EE")
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 160
    # lane-packed short-read kernel: one v_max3_i32 per cell, LDS line stage, one 16-byte store per lane and step
    _, body = pick("k_linear_lanesILi8ELb0ELb1EE")
    for ins in ("v_max3_i32", "ds_write_b128", "ds_read_b128", "global_store_dwordx4", "v_mov_b32_dpp"):
        assert ins in body, ins
    # affine and band kernels
    _, body = pick("k_affine_fillILi8ELb1EE")
    assert "v_max3_i32" in body and "global_store_dwordx4" in body
    _, body = pick("k_banded_fill_pkILi2ELb1EE")
    assert "v_pk_max_i16" in body and "v_mov_b32_dpp" in body
    assert "v_mfma" not in isa  # integer max-plus recurrence: no matrix cores anywhere
