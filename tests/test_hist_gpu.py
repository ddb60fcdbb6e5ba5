"""Far histories and the third pass of the banded kernel's hand-scheduled loop (tools/gen_hot_asm.py: hist_tail, third_pass;
dp_abi.hip: plan_far_hist) against the oracle: banded jobs whose plan serves far sites from history lines and keeps three-edge
sites in the lanes -- asserted on the plan, so that the comparison is about those paths.  pagan_batch_fetch also re-evaluates
every cell from its stored predecessors (PG_FLAG_SCORE_CHECK): a wrong cell anywhere in the band fails the call."""
import numpy as np
import pytest

import pagan2_msa_amd as pg
from pagan2_msa_amd import abi, synth
from test_far_plan_cpu import job

pytestmark = pytest.mark.gpu


def same(got, want, what):
    assert got.same_alignment(want), what


@pytest.mark.parametrize("seed", range(8))
def test_far_sites_read_history_lines(pg, oracle, seed):
    left, right, band = job(seed, n=2000 + 150 * seed, max_span=int([14, 18, 25, 30, 40, 44, 60, 22][seed]))
    n, hfl, hfr, hb, cls = pg.debug_far(left, right, band)
    assert n > 0 and ((hb & 1) & (cls == 1)).sum() > 200, "far sites on class 1 diagonals: the lanes' own far blocks run"
    model = synth.random_model(15, seed)
    flags = [0, 0, abi.OPT_NO_TERMINAL_EDGES, abi.OPT_NO_REDUCED_TERMINAL_PEN][seed % 4]
    same(pg.align(left, right, model, band, flags=flags), oracle.dp_align(left, right, model, band, flags=flags), "seed %d" % seed)


@pytest.mark.parametrize("seed", range(6))
def test_three_edge_sites_in_the_lanes(pg, oracle, seed):
    left, right, band = job(40 + seed, n=1800, p_extra=0.05, max_deg=4, max_span=int([5, 9, 12, 12, 11, 8][seed]))
    n, hfl, hfr, hb, cls = pg.debug_far(left, right, band)
    assert ((hb & 2) > 0).sum() > 200, "three-edge sites on class 1 diagonals: the third pass runs"
    model = synth.random_model(15, 50 + seed)
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), "seed %d" % seed)


@pytest.mark.parametrize("env", [{"PAGAN_DP_HIST": "0"}, {"PAGAN_DP_THREE": "0"}, {"PAGAN_DP_HIST": "0", "PAGAN_DP_THREE": "0"}])
def test_switches(pg, oracle, monkeypatch, env):
    """the A/B switches give the same alignment through the assist waves"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    left, right, band = job(5, n=1500, p_extra=0.05, max_deg=4, max_span=25)
    model = synth.random_model(15, 77)
    same(pg.align(left, right, model, band), oracle.dp_align(left, right, model, band), str(env))


def test_a_batch_of_banded_jobs_with_far_and_three_edge_sites(pg, oracle):
    jobs = []
    for seed in range(6):
        left, right, band = job(60 + seed, n=900 + 100 * seed, p_extra=0.06, max_deg=4, max_span=int([8, 16, 24, 33, 41, 12][seed]))
        jobs.append((left, right, synth.random_model(15, 60 + seed), band))
    got = pg.align_batch(jobs)
    for k, (g, j) in enumerate(zip(got, jobs)):
        same(g, oracle.dp_align(*j), "job %d" % k)
