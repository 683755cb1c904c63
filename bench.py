#!/usr/bin/env python3
"""bench.py - particle-steps/s of the SPH hot path on MI355X.

A "step" is one pass of the hot path (cell sort -> exact kNN -> density / pressure-gradient /
viscosity sums -> dt -> leapfrog) over the whole particle set, state resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles PER_GPU] [--workload NAME]

N > 1 is launched by torch.distributed.run (one rank per GPU); see sph_code_amd/multigpu.py.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# What the search kernels are bound by is vector-instruction ISSUE, and the instructions they are made of - v_pk_fma_f32 /
# v_pk_add_f32 (phase A), v_min_u32 / v_max_u32 (the ordering network), v_alignbit - each hold a SIMD for 4 cycles
# (measured: tools/valurate.hip, profiles/r03_valu_rates.txt; only the plain f32 / add forms issue in 2).  Peak in those
# units: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles.
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4.0          # = 614.4 G wave-instructions/s
# algorithmic bytes per particle-step, K = 40, int32 indices internally
B_SEARCH = 24 + 4 * 40 + 8       # R pos, W idx, W h                       = 192
B_STEP_CORE = 1248               # search + 3 passes + integrator
B_STEP_SPECIES = 1488            # + species pass (R f_un 120, W F 120)


def workload_tag(workload, forms, species, drag):
    return workload + ("_loop" if forms == "loop" else "") + ("_species" if species else "") + ("_drag" if drag else "")


def search_profile(tag, n, k):
    """Counters of the search launches from the committed rocprofv3 --pmc profile of THIS workload's command
    (profiles/latest_<tag>_profile.json, written by tools/make_search_profile.py from a tools/pmc_profile.sh run: HBM-side
    bytes, VALU issue fraction, instructions per query), or {} when there is none for this workload and size.
    These are PROFILED numbers of the same build, not measured in this run - labelled so in the bench line."""
    names = ["latest_%s_profile.json" % tag] + (["latest_search_profile.json"] if tag == "polytrope" else [])
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if int(t["n"]) == int(n) and int(t["k"]) == int(k):
                return t
        except Exception:
            pass
    return {}


def cpu_baseline(workload, n_cpu, k):
    """The oracle (NumPy/SciPy restatement of the reference, oracle/sph_oracle.py) timed on this box's host cores on
    bounded samples: one step of the bench workload at n_cpu particles on 1 thread (the headline: the reference is
    single-threaded Python), the same with the tree query on every core, and BASELINE configs[0] (1e4 particles,
    10 steps)."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    s = ics.WORKLOADS[workload](n_cpu)
    t0 = time.perf_counter()
    orc.step(s, n_neigh=k, eps=0.1, first=True, workers=1)      # eps=0.1: the reference's call
    dt = time.perf_counter() - t0
    out = {"value": n_cpu / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
           "sample": "1 step of the %s workload at N=%d (NumPy/SciPy oracle, 1 thread, eps=0.1 cKDTree "
                     "as the reference calls it); host has %d cores" % (workload, n_cpu, os.cpu_count()),
           "seconds": dt}
    t0 = time.perf_counter()
    orc.step(s, n_neigh=k, eps=0.1, first=True, workers=-1)
    dt2 = time.perf_counter() - t0
    c1 = ics.uniform_sphere(10000)
    t0 = time.perf_counter()
    for it in range(10):
        c1 = orc.step(c1, n_neigh=k, eps=0.1, first=(it == 0), workers=1)
    dt3 = time.perf_counter() - t0
    out["also"] = [
        {"value": n_cpu / dt2, "unit": "particle-steps/s", "cores": os.cpu_count(), "seconds": dt2,
         "sample": "the same step with cKDTree.query(workers=-1): the tree query on every core, the NumPy sums on one"},
        {"value": 1e4 * 10 / dt3, "unit": "particle-steps/s", "cores": 1, "seconds": dt3,
         "sample": "BASELINE configs[0]: 1e4-particle uniform sphere, 10 leapfrog steps, 1 thread"}]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=1_000_000, help="particles per GPU")
    ap.add_argument("--workload", default=None,
                    help="default: polytrope at 1 GPU (BASELINE configs[1]); uniform_cube in loop-form mode at N > 1 GPUs "
                         "(configs[3]: the reference's own IC and its time loop's own sums, natural density)")
    ap.add_argument("--neigh", dest="k", type=int, default=40, help="N_NEIGH")
    ap.add_argument("--cpu-particles", dest="cpu_n", type=int, default=600_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--device-warmup", type=int, default=3,
                    help="runs of the same W + K steps on throw-away copies of the workload before the measured run "
                         "(clock ramp of a fresh box); 0: none")
    ap.add_argument("--dt", default="reference", choices=["reference", "cfl"],
                    help="time step: the reference's rule (drv:222-229; the BASELINE metric) or a fixed Courant-"
                         "limited step (ics.cfl_dt) under which the dense 1e6 workloads stay stable")
    ap.add_argument("--forms", default=None, choices=["hydro_update", "loop"],
                    help="sums of the step: nsc.hydro_update's (the BASELINE metric) or the loop forms of the reference's "
                         "time loop (sphx_state_set_loop_forms; d set so that h(m) ~ the median kNN radius)")
    ap.add_argument("--clip-grad", action="store_true",
                    help="physics option sphx_set_clip_grad (not the reference's hydro_update arithmetic; same cost)")
    ap.add_argument("--natural-size", action="store_true",
                    help="do not enlarge the cloud beyond 1e6 particles (ics.bench_size_scale)")
    ap.add_argument("--species", action="store_true",
                    help="carry f_un and run the species pass (nsc:624-627) in every step, with the per-particle "
                         "metallicity and the AGB dust yields fused in (BASELINE configs[4]; 1488-B model)")
    ap.add_argument("--drag", action="store_true", help="gas-dust drag in the step (nsc.net_impulse, drv:455,462-463,473)")
    ap.add_argument("--gravity", default=None, choices=["direct", "tree"],
                    help="self-gravity in the timed step (off in the BASELINE metric; DESIGN 5.7)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and os.environ.get("SPHX_FORCE_DIST") != "1":
        # one process per GPU: start the ranks as children (never re-exec a process that may have touched the GPU)
        import subprocess
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    multi = world > 1 or os.environ.get("SPHX_FORCE_DIST") == "1"
    if args.workload is None:
        args.workload = "uniform_cube" if world > 1 else "polytrope"
        if args.forms is None and world > 1:
            args.forms = "loop"
    if args.forms is None:
        args.forms = "hydro_update"
    if multi:                                                     # (SPHX_FORCE_DIST=1: 1-rank rehearsal of the RCCL path)
        from sph_code_amd import multigpu
        return multigpu.bench_main(args, rank, local_rank, world)

    import torch
    import sph_code_amd.ics as ics
    from sph_code_amd.sim import Simulation

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsphx has no CPU path)")
    torch.cuda.set_device(local_rank)
    # 1 at the BASELINE size; see ics.bench_size_scale
    scale = (args.n / 1e6) ** (1. / 3.) if (args.natural_size and args.n > 1e6) else ics.bench_size_scale(args.n)
    state = ics.WORKLOADS[args.workload](args.n, size_scale=scale)
    d_loop = ics.loop_d(state, args.k) if args.forms == "loop" else None
    agb_table = None
    if args.species:
        import sph_code_amd.agb as agb_mod
        gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "agb_reference.npz")
        z = np.load(gold)                                    # the reference's AGB tables as fitted knots (data)
        agb_table = agb_mod.splines_from_arrays(z["tx"], z["ty"], z["coeffs"], z["mapto"], float(z["divisor"]))
        if state.get("f_un") is None:
            raise SystemExit("--species needs a workload that carries f_un (two_phase, dusty_sphere)")
    fixed_dt = ics.cfl_dt(state, args.k) if args.dt == "cfl" else 0.0
    # Device warm-up, apart from the W warm-up steps of the measured run: the same W + K steps on throw-away copies of the
    # workload, so that a fresh box (clocks at idle, memory pools cold) times the kernels and not its own waking up.  (The
    # same few steps again rather than one long run: under the reference's dt rule the polytrope itself stays sane for ~40
    # steps)  The measured run starts from the initial condition again.
    for _ in range(max(args.device_warmup, 0)):
        scratch = Simulation(state, n_neigh=args.k, device=local_rank, gravity=args.gravity, clip_grad=args.clip_grad,
                             forms=args.forms, d=d_loop, with_species=args.species, agb=agb_table, with_drag=args.drag)
        scratch.step(args.warmup + args.steps, fixed_dt=fixed_dt)
        torch.cuda.synchronize()
        scratch.ctx.close()
        del scratch
    sim = Simulation(state, n_neigh=args.k, device=local_rank, gravity=args.gravity, clip_grad=args.clip_grad,
                     forms=args.forms, d=d_loop, with_species=args.species, agb=agb_table, with_drag=args.drag)
    sim.step(args.warmup, fixed_dt=fixed_dt)
    sim.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(args.steps, fixed_dt=fixed_dt)      # returns after the library's stream has drained
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = sim.stats()
    failures = sim.failures()
    build_info = sim.ctx.build_info()
    final = sim.download()
    vmax = float(np.sqrt((final["velocities"] ** 2).sum(axis=1)).max())
    sane = bool(np.isfinite(final["points"]).all() and np.isfinite(final["velocities"]).all())

    # per-pass times: a few further steps with one timing event per pass (off in the timed steps: an event record between
    # two dependent kernels costs the stream ~10 us); ms_grid, ms_search and ms_total are the timed steps' own
    sim.ctx.set_timing_detail(True)
    sim.reset_stats()
    sim.step(5, fixed_dt=fixed_dt)
    torch.cuda.synchronize()
    st_detail = sim.stats()
    sim.ctx.set_timing_detail(False)
    ms_step = dt / args.steps * 1e3
    value = args.n * args.steps / dt
    ms_search = st["ms_search"] / max(st["search_steps"], 1)      # HIP events on the library's stream, around the search launches
    # (of the steps that carry them: replays of a step graph - SPHX_GRAPH - do not; when every timed step was a replay the
    #  detail steps after them, which are real steps, supply the figure)
    if not ms_search > 0.0:
        ms_search = st_detail["ms_search"] / max(st_detail["search_steps"], 1)
    achieved = B_SEARCH * args.n / (ms_search * 1e-3) / 1e9 if ms_search > 0.0 else None
    tag = workload_tag(args.workload, args.forms, args.species, args.drag)
    prof = search_profile(tag, args.n, args.k)
    # the a6-inclusive figure beside the core one (nsc.hydro_update always forms F[s,i], nsc:624-627; the headline's step
    # leaves it out, as `config` says): the same workload with the species pass in the step, a short run outside the timed one
    with_species_ms = None
    if not args.species and state.get("f_un") is not None and os.environ.get("SPHX_BENCH_SPECIES_LINE", "1") != "0":
        sp = Simulation(state, n_neigh=args.k, device=local_rank, gravity=args.gravity, clip_grad=args.clip_grad,
                        forms=args.forms, d=d_loop, with_species=True, with_drag=args.drag)
        sp.step(args.warmup, fixed_dt=fixed_dt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sp.step(10, fixed_dt=fixed_dt)
        torch.cuda.synchronize()
        with_species_ms = (time.perf_counter() - t1) / 10 * 1e3
        sp.ctx.close()
        del sp
    ipq = prof.get("valu_wave_instr_per_query")
    valu_achieved = (ipq * args.n / (ms_search * 1e-3) / 1e9) if (ipq and ms_search > 0.0) else None
    b_step = B_STEP_SPECIES if args.species else B_STEP_CORE
    out = {
        "metric": "particle-steps/sec + achieved HBM GB/s, 10^6-particle 3D sphere, 1/2/4/8 GPUs",
        "value": value, "unit": "particle-steps/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "device_warmup_runs_on_scratch_copies": args.device_warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s, N=%d, K=%d, fp64, poly6 kernel, viscosity on" %
                               (args.workload, args.n, args.k),
                   "particles_per_gpu": args.n, "decomposition": "single GPU",
                   "gravity": args.gravity or "off", "dt": args.dt, "cloud_size_scale": scale,
                   "forms": args.forms, "species_pass": bool(args.species), "drag": bool(args.drag),
                   "kernel_gradient": "clipped (physics option)" if args.clip_grad else "as nsc.hydro_update (unclipped)",
                   # the library as built and the SPHX_* environment of this run: the configuration the parity tests ran is
                   # "experiments=0" with no SPHX_* variable set
                   "build": build_info,
                   "env": {k_: v for k_, v in sorted(os.environ.items()) if k_.startswith("SPHX_")}},
        "with_species_ms_per_step": with_species_ms,
        # the reference's scheme can diverge: evidence that the timed window was a sane flow
        "state_check": {"finite": sane, "max_speed_m_s": vmax,
                        "max_drift_per_step_in_mean_h": vmax * float(final["dt"]) / float(final["sizes"].mean()),
                        # the library's own failure counters over the timed steps (sphx_stats.bad_*: by ballot inside the
                        # update kernels - what the reference's nan_to_num guards hid; no download needed)
                        "failures": failures},
        # The dominant kernel is the search (knn_group_kernel + the list-mode knn_kernel for what it hands on).  SURVEY
        # 8(d) prices it against HBM (achieved / peak / frac below, live from this run's events); the counters say it
        # is bound by instruction issue and latency, not by HBM: `limiter` and the profiled fractions name that.
        # ONE bound named: vector-instruction issue.  achieved = wave-instructions the search's two launches issue per second
        # (instructions per query from the committed counter profile of this build and workload x the queries of this
        # run / the launches' time measured in THIS run with HIP events on the library's stream); peak as above; frac =
        # achieved / peak.  The counter view of the same thing (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles at the clock the
        # profiled run held) rides along as valu_issue_frac_counters.  The HBM pricing SURVEY 8(d) asks for - algorithmic
        # 192 B per query against the 8 TB/s peak, and the counter traffic - is the `hbm` object: a few per cent, which is
        # the point: this path is not memory-bound.
        "roofline": {"bound": "valu_issue", "kernel": "search: knn_group_kernel + knn_kernel<0,1,1> (list mode)",
                     "achieved": valu_achieved, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                     "frac": (valu_achieved / VALU_PEAK_GINSTR) if valu_achieved else None,
                     "traffic": prof.get("traffic_bytes_per_launch"),
                     "traffic_source": prof.get("source", None),
                     "valu_issue_frac_counters": prof.get("valu_issue_frac"),
                     "valu_wave_instr_per_query": ipq,
                     "l2_hit_rate": prof.get("l2_hit_rate"),
                     "profiled_counters_source": prof.get("source", None),
                     "kernel_ms": ms_search,
                     "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                             "algorithmic_bytes_per_launch": B_SEARCH * args.n, "algorithmic_bytes_per_particle": B_SEARCH,
                             "traffic": prof.get("traffic_bytes_per_launch")}},
        "step_model": {"algorithmic_bytes_per_particle_step": b_step,
                       "achieved_GBs": b_step * value / 1e9,
                       "frac_of_hbm_peak": b_step * value / 1e9 / HBM_PEAK_GBS},
        "per_pass_ms": dict({"ms_search": ms_search, "ms_total": st["ms_total"] / max(st["steps"], 1)},
                            **{k_: st_detail[k_] / max(st_detail["detail_steps"], 1) for k_ in
                               ("ms_grid", "ms_prep", "ms_density", "ms_species", "ms_pi", "ms_visc", "ms_integrate", "ms_gravity")},
                            source="ms_search / ms_total: the timed steps (events around the search's launches and around "
                                   "the whole call); the others: 5 further steps with one timing event per pass "
                                   "(sphx_set_timing_detail)"),
        "search": {"candidates_per_particle_step": st["candidates"] / max(st["steps"], 1) / args.n,
                   "retries_per_step": st["retries"] / max(st["steps"], 1), "cells": st["cells"],
                   "fallback_queries_last_step": st.get("fallback_queries", 0), "short_rows": st.get("short_rows", 0), "far_queries": st.get("far_queries", 0),
                   "outlier_levels": st.get("outlier_levels", 0),
                   "refresh_steps": st["refresh_steps"], "rebuild_steps": st["rebuild_steps"],
                   "graph_steps": st.get("graph_steps", 0)},
    }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_n, args.k)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
