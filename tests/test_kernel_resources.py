"""Register budget of the hot kernels, read from the code objects inside libbkcore.so (no GPU needed).  The headline
forward runs three workgroups per CU only at <= 168 vector registers, and both it and the one-pass analytic backward
lose a large factor when the compiler starts spilling in their chunk loops (measured: 0.18 -> 0.23 ms for 48 spilled
dwords in the forward, 0.48 -> 0.78 ms in the analytic kernel) -- a change that does that should fail here, not in a
benchmark."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "project-resnet-bk-an-o-n-language-model-architecture_amd", "bkcore", "libbkcore.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(LIB) and os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    d = tmp_path_factory.mktemp("codeobj")
    shutil.copy(LIB, d / "libbkcore.so")
    subprocess.run([objdump, "--offloading", "libbkcore.so"], cwd=d, check=True, capture_output=True)
    out = {}   # (code object, kernel) -> resources: translation units may hold kernels of the same name
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        txt = subprocess.run([readelf, "--notes", f], cwd=d, check=True, capture_output=True, text=True).stdout
        name = None
        for line in txt.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = f + ":" + m.group(1)
                out[name] = {}
                continue
            m = re.match(r"\s+\.(private_segment_fixed_size|vgpr_count|group_segment_fixed_size):\s+(\d+)", line)
            if m and name:
                out[name][m.group(1)] = int(m.group(2))
    assert out
    return out


def _find(kernels, *parts):
    hits = [k for k in kernels if all(p in k for p in parts)]
    assert hits, parts
    return hits


def test_headline_forward_fits_three_workgroups_per_cu(kernels):
    # bk_ratio_forward_kernel<float, PHI_EXACT, MODE 0, SAVE false, BST false> of the float32 translation unit (the one
    # for rows longer than 49152, with its chains in double, has the same names and a two-workgroup budget)
    hits = _find(kernels, "bk_ratio_forward_kernelIfLb", "ELi0ELb0ELb0E")
    by_unit = {}
    for k in hits:
        by_unit.setdefault(k.split(":")[0], []).append(kernels[k])
    units = [u for u, rs in by_unit.items() if all(r["vgpr_count"] <= 168 for r in rs)]
    assert len(units) == 1, by_unit
    for r in by_unit[units[0]]:
        assert r["private_segment_fixed_size"] <= 16, r               # at most a few loop-invariant dwords
        assert 3 * r["group_segment_fixed_size"] <= 160 * 1024, r


def test_forward_in_units_keeps_the_same_budget(kernels):
    # bk_ratio_forward_kernel<float, PHI_EXACT, MODE 3, SAVE false>: rows cut into units of one launch run the chunk
    # loop of MODE 0; a formulation that keeps one more scalar alive across that loop spills 30-100 dwords
    hits = _find(kernels, "bk_ratio_forward_kernelIfLb", "ELi3ELb0ELb0E")
    assert len(hits) == 2, hits
    for k in hits:
        r = kernels[k]
        assert r["vgpr_count"] <= 168 and r["private_segment_fixed_size"] <= 16, (k, r)
        assert 3 * r["group_segment_fixed_size"] <= 160 * 1024, (k, r)


def test_forward_in_chunks_keeps_the_same_budget(kernels):
    # bk_ratio_forward_kernel<float, PHI_EXACT, MODE 4, SAVE false, BST false>: one workgroup per chunk with the
    # look-back hand-over; three per CU with float32 chains, two per CU (256 registers) with double chains
    hits = _find(kernels, "bk_ratio_forward_kernelIfLb", "ELi4ELb0ELb0E")
    assert len(hits) == 4, hits
    small = [k for k in hits if kernels[k]["vgpr_count"] <= 168]
    assert len(small) == 2, hits
    for k in hits:
        r = kernels[k]
        per_cu = 3 if k in small else 2
        assert r["private_segment_fixed_size"] == 0, (k, r)
        assert per_cu * r["group_segment_fixed_size"] <= 160 * 1024, (k, r)


def test_forward_keeping_boundary_states_keeps_the_same_budget(kernels):
    # bk_ratio_forward_kernel<float, PHI_EXACT, MODE 0 / 3 / 4, SAVE false, BST true>: the variant that also stores the
    # values its chains carry into every block of 8 elements (each stored the moment it exists, at an address worked
    # out on the spot) runs three workgroups per CU like the plain one
    hits = _find(kernels, "bk_ratio_forward_kernelIfLb", "ELb0ELb1E")
    assert len(hits) == 6, hits                                       # MODE 0, 3 and 4, both phi conventions
    for k in hits:
        r = kernels[k]
        assert r["vgpr_count"] <= 168 and r["private_segment_fixed_size"] <= 16, (k, r)
        assert 3 * r["group_segment_fixed_size"] <= 160 * 1024, (k, r)


def test_one_pass_analytic_backward_does_not_spill(kernels):
    for k in _find(kernels, "bk_adjoint_sweeps"):
        r = kernels[k]
        assert r["private_segment_fixed_size"] == 0, (k, r)
        assert r["vgpr_count"] <= 256 and 2 * r["group_segment_fixed_size"] <= 160 * 1024, (k, r)


def test_streaming_kernels_do_not_spill(kernels):
    for part in ("bk_backward_ref_kernel", "bk_layer_backward_ref_kernel", "bk_layer_prologue_kernel",
                 "bk_cumsum_row_kernel", "bk_cumsum_strided_kernel"):
        for k in _find(kernels, part):
            assert kernels[k]["private_segment_fixed_size"] == 0, (k, kernels[k])
