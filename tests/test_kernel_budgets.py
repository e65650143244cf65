"""Compile-time budgets of the hot kernels, checked without a GPU: hipcc cross-compiles gfx950 here, `-Rpass-analysis=kernel-resource-usage`
reports registers / scratch / occupancy per kernel, and the device assembly shows whether a gather was issued in one piece.  These are
the properties the measured numbers rest on; a change that silently spills the matcher's sweep, drops its
occupancy, or lets the compiler serialise the re-examination's loads again (round 3: ten memory round trips per step instead of one,
profiles/r03_match_phases_before.txt) fails here instead of showing up as a slower bench line."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opensfm_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage"]  # the product's flags (csrc/build.sh) + assembly + remarks

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def compile_device(name, tmp_path_factory, cache={}):
    """(assembly text, {mangled kernel name: {field: int}}) of csrc/<name>.hip"""
    if name in cache:
        return cache[name]
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    r = subprocess.run([HIPCC, *FLAGS, os.path.join(CSRC, name + ".hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    cache[name] = (out.read_text(), kernels)
    return cache[name]


def one(kernels, *parts):
    hits = [k for k in kernels if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return kernels[hits[0]], hits[0]


def body(asm, mangled):
    a = asm.index("\n" + mangled + ":")
    return asm[a: asm.index("s_endpgm", a)]


def test_matcher_sweep_keeps_its_registers_and_two_workgroups_per_cu(tmp_path_factory):
    asm, k = compile_device("match", tmp_path_factory)
    for fq in ("ILb0E", "ILb1E"):  # integer store / float store (FQ)
        r, name = one(k, "match_fused_kernel", fq)
        assert r["Occupancy"] == 2 and r["AGPRs"] == 0  # __launch_bounds__(256, 2): two workgroups of 78 KiB LDS per CU
        assert r["ScratchSize"] <= 128 and r["VGPRs Spill"] <= 32, r  # a handful of values parked around the chunk boundary, nothing more
        text = body(asm, name)
        # no scratch traffic and no full wait inside a basic block that issues MFMAs (the sweep)
        for block in re.split(r"\n\.LBB\d+_\d+:", text):
            if "v_mfma_i32_32x32x32_i8" in block:
                assert "scratch_" not in block
        assert text.count("v_mfma_i32_32x32x32_i8") >= 2 * 64  # two instantiations of the 64-MFMA step (plain / gathered queries)


def test_reexamination_issues_its_gather_in_one_piece(tmp_path_factory):
    """the class re-examination: 8 query slices + 2 x 8 target slices + 2 norms per lane, all in flight before the first dot product"""
    asm, k = compile_device("match", tmp_path_factory)
    _, name = one(k, "match_fused_kernel", "ILb0E")
    lines = [ln.strip() for ln in body(asm, name).splitlines() if ln.strip() and not ln.strip().startswith(";")]
    dots = [i for i, ln in enumerate(lines) if ln.startswith("v_dot4")]
    # the first long run of dot products is the re-examination of pass A (64 per step)
    start = next(i for i in dots if sum(1 for j in dots if i <= j < i + 90) >= 60)
    window = lines[max(0, start - 110): start]
    loads = [i for i, ln in enumerate(window) if ln.startswith("global_load_dwordx4")]
    assert len(loads) >= 24, len(loads)
    between = window[loads[0]: loads[-1]]
    assert not any(ln.startswith("s_waitcnt vmcnt(0)") for ln in between)  # was: one full wait per query slice
    assert not any(ln.startswith("s_cbranch") for ln in between)            # and a branch around every target load


def test_ba_streaming_kernels_do_not_spill(tmp_path_factory):
    _, k = compile_device("ba", tmp_path_factory)
    # (length-prefixed as in the mangled names: "17schur_shot_kernel" is not "21gen_schur_shot_kernel")
    for parts in (("schur_point_coop_kernel", "ILi0E"), ("17schur_shot_kernel", "ILi1E"), ("17schur_shot_kernel", "ILi4E"), ("11eval_kernel", "ILb1E"), ("11eval_kernel", "ILb0E"),
                  ("band_assemble_kernel",), ("19border_point_kernel", "ILi3E"), ("18border_shot_kernel", "ILi3ELi1E"), ("18border_shot_kernel", "ILi3ELi4E"), ("19precond_shot_kernel",),
                  ("16shot_grad_kernel", "ILi1E"), ("16shot_grad_kernel", "ILi4E"), ("bcr_level_kernel", "ILi9E"), ("wide_factor_kernel",), ("wide_push_kernel", "ILi1ELb0E")):
        r, name = one(k, *parts)
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (name, r)
    r, _ = one(k, "schur_point_coop_kernel", "ILi0E")
    assert r["Occupancy"] >= 4  # the mat-vec is a streaming kernel: it needs the waves to cover HBM latency
    # round 6: pass B of the mat-vec recomputes its Jacobian rows (sm_row: ~400 fp64 operations per observation instead of 160 bytes read); it must
    # keep the shot's frame in scalar registers and three waves per SIMD to cover the gathers of w and of the points
    r, _ = one(k, "17schur_shot_kernel", "ILi1E")
    assert r["Occupancy"] >= 3, r
    r, _ = one(k, "bcr_level_kernel", "ILi9E")
    assert r["LDS Size"] <= 160 * 1024
    # the generic mode's streaming kernels (ba_generic.inc): the mat-vec pair without scratch at every border width, pass A at streaming occupancy
    for nr in (2, 3):
        for mode in (0, 1, 2):
            # (round 6: modes 0 and 2 of the two-row kernel also come specialised per projection type for the COMPACT rows, whose border slots they
            #  rebuild with project_full: the same budget)
            hits = [n for n in k if "gen_schur_point_kernelILi%dELi%dE" % (nr, mode) in n]
            assert len(hits) == (4 if nr == 2 and mode != 1 else 1), hits
            for name in hits:
                r = k[name]
                compact = name.endswith("ELb1EEEvNS_3DevEPKd")  # (project_full inlined: the fisheye_opencv form takes 160 registers, three waves per SIMD)
                assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0 and r["Occupancy"] >= (3 if compact else 4), (name, r)
    # round 6: the generic per-instance kernels recompute their rows too (gen_sm_row).  Specialised for the projection type every camera has (BROWN 2,
    # FISHEYE_OPENCV 3, PERSPECTIVE 0) they use no scratch memory; the unspecialised ones (mixed models, spherical) index the parameter
    # Jacobian dynamically and do -- the price of the evaluation kernel's unspecialised form as well
    for model in (0, 2, 3):
        for kw in (4, 9, 16, 22):
            for kern in ("gen_schur_shot_kernel", "gen_shot_grad_kernel"):
                r, name = one(k, kern, "ILi2ELi%dELi%dE" % (kw, model))
                assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (name, r)
    r, name = one(k, "gen_schur_shot_kernel", "ILi2ELi9ELi2E")  # a Brown camera's nine columns: two waves per SIMD
    assert r["Occupancy"] >= 2, (name, r)


def test_hahog_per_feature_kernels(tmp_path_factory):
    _, k = compile_device("hahog", tmp_path_factory)
    r, _ = one(k, "orientation_kernel")
    assert r["ScratchSize"] == 0 and r["LDS Size"] <= 40 * 1024  # four workgroups per CU (round 4: the patch shares the records' space)
    assert r["Occupancy"] >= 4 and r["VGPRs"] <= 128  # four pixels of the resampling in flight, not seven (149 registers: three waves per SIMD)
    r, _ = one(k, "descriptor_kernel")
    assert r["ScratchSize"] == 0 and r["LDS Size"] <= 26 * 1024  # six
    assert r["Occupancy"] >= 6 and r["VGPRs"] <= 84  # two pixels of the resampling in flight, not four (90 registers: five waves)


def test_hahog_extremum_search_holds_its_rows_in_registers(tmp_path_factory):
    """round 6: the extremum search takes the ten rows of the five levels into registers (no scratch), gets the columns beside a lane by DPP
    wave shifts (no LDS traffic for them) and keeps the candidate list of a tile within 16 KB of LDS"""
    asm, k = compile_device("hahog", tmp_path_factory)
    r, name = one(k, "extrema_kernel")
    assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0 and r["LDS Size"] <= 16 * 1024 and r["Occupancy"] >= 3, r
    b = body(asm, name)
    assert b.count("wave_shr:1") == 50 and b.count("wave_shl:1") == 50, (b.count("wave_shr:1"), b.count("wave_shl:1"))  # 5 levels x 10 rows, both sides
    assert "ds_bpermute" not in b


def test_hahog_fused_smoothing_and_wide_band_kernels(tmp_path_factory):
    """round 4: the fused separable smoothing keeps its sliding windows in registers (no scratch) and at least four workgroups per CU for
    every tap count; the wide band's hand-written kernels do not spill"""
    _, k = compile_device("hahog", tmp_path_factory)
    for W in (1, 4, 5, 6, 8, 10, 16):
        r, _ = one(k, "smooth_fused_kernel", "ILi%dE" % W)
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0 and r["LDS Size"] <= 40 * 1024 and r["Occupancy"] >= 4, (W, r)
    _, k = compile_device("ba", tmp_path_factory)
    for parts in (("dbcr_sweep_kernel", "ILi4E"), ("dbcr_sweep_kernel", "ILi1E"), ("dgj_pivot_kernel",), ("dbcr_transpose_kernel",), ("band_mfma_kernel", "ILi4E")):
        r, _ = one(k, *parts)
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (parts, r)
