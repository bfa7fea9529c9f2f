"""Regression guard on the compiled recurrence kernels (no GPU: hipcc cross-compiles).

The hot kernels sit right at register-file steps (128 VGPRs = 4 waves/SIMD, 168 = 3); an edit
that adds a few live registers silently costs an occupancy level or spills, which measured
10-30 % on the MI355X.  This test pins what the measured numbers rely on.
"""

import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def resources():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    import kernel_resources

    return kernel_resources.collect()


def _row(resources, name):
    # (template instances demangle with their return type, plain kernels without)
    matches = [row for key, row in resources.items() if key.startswith(f"void bdg::{name}(") or key.startswith(f"bdg::{name}(")]
    assert len(matches) == 1, (name, [k for k in resources if name.split("<")[0] in k][:5])
    return matches[0]


@pytest.mark.timeout(600)
def test_headline_kernels_do_not_spill_and_keep_their_occupancy(resources):
    headline = _row(resources, "cheb_step_dict<bdg::RealPHMode, 4, 5, false>")
    assert headline["scratch"] == 0 and headline["vgpr"] <= 128 and headline["occupancy"] >= 4
    for mode, lanes in (("RealPHMode", 4), ("ComplexPHMode", 8), ("RealMode", 4), ("ComplexMode", 8)):
        row = _row(resources, f"cheb_step_pipelined<bdg::{mode}, {lanes}, 5>")
        assert row["scratch"] == 0 and row["vgpr"] <= 168 and row["occupancy"] >= 3, (mode, row)
    for lanes in (4, 8, 16, 32, 64):
        row = _row(resources, f"cheb_step<bdg::ComplexMode, {lanes}, false>")
        assert row["scratch"] == 0 and row["occupancy"] >= 4, (lanes, row)


@pytest.mark.timeout(600)
def test_sweep_kernels_fit_two_waves_per_simd_without_spilling(resources):
    """K7 / K7b / K8 keep five to seven site arrays and a plane of prefetch in registers: they must stay
    within the 256-VGPR step (two waves per SIMD) and must not spill, in every arithmetic mode, lane
    count and marching direction."""
    for mode in ("RealPHMode", "ComplexPHMode", "RealMode", "ComplexMode"):
        for reverse in ("false", "true"):
            for lanes in (1, 2, 4):
                row = _row(resources, f"cheb_sweep<bdg::{mode}, {lanes}, {reverse}>")
                assert row["scratch"] == 0 and row["vgpr"] <= 256 and row["occupancy"] >= 2, (mode, lanes, reverse, row)
            for lanes in (2, 4):
                for gen in ("false", "true") if reverse == "false" else ("false",):  # (GEN: start block made in registers)
                    row = _row(resources, f"cheb_sweep3<bdg::{mode}, {lanes}, {reverse}, {gen}, 0, 4>")
                    assert row["scratch"] == 0 and row["vgpr"] <= 256 and row["occupancy"] >= 2, (mode, lanes, reverse, gen, row)
        for lanes in (2, 4):
            for nt in ("false", "true"):  # (non-temporal t_{n-1} loads / t_{n+1} stores: a compiled form of its own since round 4)
                row = _row(resources, f"cheb_roll3<bdg::{mode}, {lanes}, {nt}>")
                assert row["scratch"] == 0 and row["vgpr"] <= 256 and row["occupancy"] >= 2, (mode, lanes, nt, row)


@pytest.mark.timeout(600)
def test_streamed_onsite_sweep_keeps_two_waves_per_simd(resources):
    """cheb_sweep3<..., OS, WAVES> carries the prefetched records on top of K7b's state.  It must keep two waves per SIMD and -
    since round 4 - must not touch scratch at all: a reload of a spilled value waits with vmcnt(0), i.e. for every prefetched
    plane in flight (that cost the complex form a quarter of its launch time until buffer addressing freed the registers;
    DESIGN §4).  Also the forms with 2 lanes per site and the complex site records, in workgroups of seven waves."""
    forms = [("RealPHMode", 4, 1, 4, 0), ("ComplexPHMode", 4, 1, 4, 0), ("RealPHMode", 4, 2, 4, 0),
             ("RealPHMode", 2, 1, 4, 0), ("RealPHMode", 2, 1, 7, 0), ("ComplexPHMode", 2, 1, 7, 0), ("ComplexPHMode", 4, 2, 7, 0)]
    for mode, lanes, streamed, waves, limit in forms:
        for reverse, gen in (("false", "false"), ("true", "false"), ("false", "true")):
            row = _row(resources, f"cheb_sweep3<bdg::{mode}, {lanes}, {reverse}, {gen}, {streamed}, {waves}>")
            # (the form that makes the start block in registers runs the first sweep of a run only - 1 launch in 21: a few spills allowed)
            allowed = 64 if gen == "true" else limit
            assert row["scratch"] <= allowed and row["vgpr"] <= 256 and row["occupancy"] >= 2, (mode, lanes, streamed, waves, reverse, gen, row)


@pytest.mark.timeout(600)
def test_chunk_kernel_and_dense_two_stage_kernels_keep_their_registers(resources):
    """Round 4: cheb_march3 (the sweeps of a chunk in one launch) carries the unit body of cheb_sweep3 twice (with and
    without the generated start block) plus its claim / wait code: two waves per SIMD, a few loop-invariant values in
    scratch at most.  The MFMA kernels of the two-stage dense route (csrc/twostage.hpp) must not spill at all; the bulge
    chasing and the band inverse iteration keep their blocks / window in registers (one wave per workgroup: the budget
    is the whole file), again without scratch."""
    for mode in ("RealPHMode", "ComplexPHMode", "RealMode", "ComplexMode"):
        for lanes in (2, 4):
            row = _row(resources, f"cheb_march3<bdg::{mode}, {lanes}, 0>")
            assert row["scratch"] <= 96 and row["vgpr"] <= 256 and row["occupancy"] >= 2, (mode, lanes, row)
    for name in ("ts_symm", "ts_rank2k", "ts_xz", "ts_gram", "ts_vgram2", "ts_vtz", "ts_zupdate", "ts_qr_apply", "ts_qr_verify"):
        row = _row(resources, name)
        assert row["scratch"] == 0, (name, row)
    for name in ("ts_chase", "ts_band_vectors"):
        row = _row(resources, name)
        assert row["scratch"] == 0 and row["vgpr"] <= 256, (name, row)
