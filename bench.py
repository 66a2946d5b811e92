#!/usr/bin/env python
"""Headline benchmark: restarts x coord-sweeps / sec of improve(COORD_DESCENT) on Boolean least
squares n=1024, m=256 (rows of A), 4096 random restarts per GPU (BASELINE.json configs[1]).

One "step" = suggest(RANDOM) for the whole population (device Philox) + improve_coord_descent
(phase 1 + gate + phase 2 to convergence, reference defaults) on the rank's restarts + selection of
the best (objective, max-violation) point -- everything resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1: either a launcher provides RANK / LOCAL_RANK / WORLD_SIZE (python -m torch.distributed.run ...)
or -- plain `python bench.py --gpus N` -- this process becomes rank 0 and starts ranks 1..N-1 itself
(qcqp_amd.dist.spawn_local_ranks; the RCCL unique id travels through a file rendezvous, no torch).
Weak scaling (default): every rank runs --restarts restarts with disjoint GLOBAL restart indices;
strong: --restarts restarts in total, split over the ranks.  The only collective is the best-point
selection (RCCL inside libqcqp_mi.so: two all-reduces for all steps of a streamed run).

`value` counts PHASE-2 restart-sweeps only (the unit SURVEY.md section 8d defines: 2 n^2 flops each);
phase-1 sweeps (element-wise for this family) are reported separately.

How the steps are scheduled (--scheme, DESIGN.md sections 4.1d / 6).  `stream` (default): the K timed steps are ONE persistent
launch of the lifecycle kernel per GPU (qcqpmi_cd_stream_run): a workgroup owns 16 restart slots and a slot that becomes free
draws the next restart of the run and runs its whole step itself -- keyed normals, phase 1, gate, phase 2, objective -- so no
step waits for the slowest restart of the previous one; then one selection launch per step and ONE exchange over the ranks for
all steps.  `two` (rounds 2 / 3): two contexts per GPU, one phase-2 launch per step, the preparation of step k + 1 in the tail of
launch k.  `auto`: stream, and on one GPU the same steps through `two` as well (`schemes` in the line; same best point).

Prints ONE JSON line on rank 0: a compact record (< 4 KB, strict JSON), alone on the real stdout and last -- everything else
(RCCL's banner included) goes to stderr.  `--secondary` (the other BASELINE.json configurations, ~2.5 minutes) and `--compare`
(the other scheme / the other lifecycle kernel on the same steps) write a sidecar file (gpurun_out/bench_secondary.json) instead
of growing the line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
from qcqp_amd import _threads  # noqa: E402

if '--gpus' in sys.argv[:-1]:          # the ranks of a multi-GPU run share the machine's CPU quota (this process may spawn them itself)
    try:
        os.environ.setdefault('QCQP_LOCAL_RANKS', str(max(1, int(sys.argv[sys.argv.index('--gpus') + 1]))))
    except ValueError:
        pass
_threads.set_blas_env()      # BEFORE NumPy: 256 visible CPUs, 16 cores of cgroup quota -- spinning BLAS workers get the whole container throttled

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak (AMD datasheet); see DESIGN.md section 4


def profiled_counters(kernel_name):
    """PMC-derived numbers of the headline kernel from the committed profile summary of this round
    (profiles/rNN_summary.json, written by tools/summarize_profile.py from separate rocprofv3 --pmc passes of
    THIS command line).  Not measured in this run: returned with their provenance (file, git commit of the
    profile, date, kernel, what one launch was) so that the bench line says where they come from; only a summary of the
    kernel this run reports is used; None if absent."""
    pdir = os.path.join(REPO, 'profiles')
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith('_summary.json'):
            try:
                d = json.load(open(os.path.join(pdir, name)))
            except Exception:
                continue
            exact = d.get('engine_kernel') == kernel_name
            if best is not None and best['exact'] and not exact:
                continue                 # a profile of exactly this instantiation beats a later one of the same kernel family
            if 'cd_phase2_hbm_bytes_per_launch' in d and (exact or (d.get('kernel') or '').split('<')[0] == (kernel_name or '').split('<')[0]):
                best = dict(exact=exact, traffic=d['cd_phase2_hbm_bytes_per_launch'], mfma_busy=d.get('cd_phase2_mfma_busy_frac'),
                            source='profiles/' + name, profile_commit=d.get('git_commit'), profile_date=d.get('date'),
                            kernel=d.get('kernel'), launch=d.get('launch'))
    if best is not None:
        best.pop('exact')
    return best


# ------------------------------------------------------------------------------------- the one line
HEADLINE_LIMIT = 4000       # bytes; the driver parses the LAST stdout line (BENCH_r05: a 20 KB line followed by RCCL's banner was not parsed)


def claim_stdout():
    """File descriptor 1 is kept for the ONE line: from here on everything else this process (Python or a C library -- RCCL prints
    its version banner through C stdio when the communicator is created, and stdio flushes it at exit, i.e. AFTER anything Python
    printed) writes to "stdout" lands on stderr.  Returns the descriptor the line is written to."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return keep


def _finite(o):
    """Strict JSON: non-finite floats become null, numpy scalars become Python numbers."""
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, float)):
        f = float(o)
        return f if np.isfinite(f) else None
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


def headline_line(res):
    """The compact record (< HEADLINE_LIMIT bytes, strict JSON, one line): optional keys are dropped, least important first,
    should a field ever grow -- the contract's keys never are."""
    res = _finite(res)
    optional = ['phase1', 'timed_region_note', 'sidecar']
    line = json.dumps(res, allow_nan=False, separators=(',', ':'))
    while len(line) >= HEADLINE_LIMIT and optional:
        res.pop(optional.pop(0), None)
        line = json.dumps(res, allow_nan=False, separators=(',', ':'))
    if len(line) >= HEADLINE_LIMIT:
        for key in ('roofline', 'cpu_baseline', 'config', 'best'):
            for sub in [k for k, v in res.get(key, {}).items() if isinstance(v, str) and len(v) > 80]:
                res[key][sub] = res[key][sub][:77] + '...'
        line = json.dumps(res, allow_nan=False, separators=(',', ':'))
    return line


def emit_line(fd, line):
    """The line, alone, as the last thing on the real stdout."""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # C stdio buffers (RCCL's banner) go where descriptor 1 points now: stderr
    except Exception:
        pass
    data = (line + '\n').encode()
    while data:
        data = data[os.write(fd, data):]


def write_sidecar(path, payload):
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(_finite(payload), f, indent=1, allow_nan=False)
        return True
    except Exception as ex:
        sys.stderr.write('bench: sidecar %s not written: %r\n' % (path, ex))
        return False


# ------------------------------------------------------------------------------------- secondary records
def secondary_records(device, sdr_full=False):
    """Bounded measurements of the other BASELINE.json configurations on ONE GPU (not part of `value`): the share of
    one rank, inputs resident, a few seconds in total.  Each record names its workload, kernel time and roofline."""
    import numpy as np
    from qcqp_amd import lowrank, problems
    from qcqp_amd.engine import Engine
    from qcqp_amd.form import QCQPForm
    recs = []
    # the headline family with 4 tiles per CU: the hardware's workgroup queue refills a CU as soon as its tile of 16 restarts
    # has converged, so the 1-workgroup-per-CU straggler effect of the headline (kernel time = slowest tile) is amortised
    try:
        n = 1024
        funcs, _, _ = problems.boolean_least_squares(n, 256, seed=1)
        e = Engine(QCQPForm.from_arrays(funcs), device=device)
        pts = []
        for R, reps in ((16384, 4), (65536, 2)):
            for mode in (0, 2):          # 0: tile-bound cd_phase2_q_kernel; 2 (default dispatch): slot-queue kernel when tiles > CUs
                e.cd_queue(mode)
                e.randn(R, seed=90)
                e.cd_run(phase1=True, seed=90)
                sw = ms = 0.0
                e.sync()
                t0 = time.perf_counter()
                for k in range(reps):
                    e.randn(R, seed=91 + k)
                    out = e.cd_run(phase1=True, seed=91 + k)
                    sw += float(out['visits2'].sum()) / n
                    ms += e.kernel_ms(Engine.KERNEL_CD2)
                e.sync()
                dt = time.perf_counter() - t0
                pts.append({'restarts': R, 'kernel': e.last_cd_kernel(), 'value': sw / dt, 'kernel_ms_per_launch': ms / reps,
                            'achieved': sw * 2.0 * n * n / 1e12 / (ms / 1e3), 'frac': sw * 2.0 * n * n / 1e12 / (ms / 1e3) / FP64_PEAK_TFLOPS})
        head = [p_ for p_ in pts if p_['restarts'] == 16384 and p_['kernel'] == 'cd_phase2_qs_kernel'][0]
        recs.append({'config': 'headline family (Boolean LS n=1024 m=256) with 16384 / 65536 restarts on one GPU (1024 / 4096 tiles on 256 CUs): '
                               'tile-bound kernel (a workgroup runs a tile of 16 restarts until the slowest has converged) against the '
                               'slot-queue kernel (a converged restart is replaced at the next sweep boundary); first figures: 16384 restarts, '
                               'slot queue',
                     'metric': 'restarts x coord-sweeps / s (phase 2)', 'value': head['value'], 'unit': 'restart-sweeps/s',
                     'kernel': head['kernel'], 'by_restarts_and_kernel': pts,
                     'roofline': {'bound': 'mfma', 'kernel': head['kernel'], 'achieved': head['achieved'],
                                  'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': head['frac'],
                                  'kernel_ms_per_launch': head['kernel_ms_per_launch']}})
        del e
    except Exception as ex:
        recs.append({'config': 'headline family, 16384 restarts', 'error': repr(ex)})
    # the lifecycle kernel in steady state (40 steps of 4096 restarts in one launch) and at the STRONG-SCALING share of the named
    # configuration: 4096 restarts over 8 GPUs = 512 restarts per GPU and step -- streamed, a step of 512 restarts costs an eighth of
    # a step of 4096 (the slots never wait for a step to end), where one launch per step is tile-bound (32 tiles on 256 CUs)
    try:
        n = 1024
        funcs, _, _ = problems.boolean_least_squares(n, 256, seed=1)
        e = Engine(QCQPForm.from_arrays(funcs), device=device)
        try:        # the headline's objective factor (P0 = L L^T of rank 256: the lifecycle kernel carries L^T X), as in main()
            from qcqp_amd import lowrank
            P0s = funcs[0][0]
            Lf = lowrank.objective_factor(P0s.toarray() if hasattr(P0s, 'toarray') else np.asarray(P0s), max_rank=288)
            if Lf is not None:
                e.cd_set_objective_factor(Lf)
        except Exception:
            pass
        pts = []
        stream_kernel = None
        for R, K in ((4096, 40), (512, 160)):
            e.cd_stream_run(K, R, seed=300, seed_stride=1)
            stream_kernel = e.last_cd_kernel()
            e.sync()
            t0 = time.perf_counter()
            o = e.cd_stream_run(K, R, seed=400, seed_stride=1)
            e.sync()
            dt = time.perf_counter() - t0
            ms = e.kernel_ms(Engine.KERNEL_CD2)
            sw = float(o['visits2'].sum()) / n
            pts.append({'restarts_per_step': R, 'steps': K, 'value': sw / dt, 'ms_per_step': 1e3 * dt / K, 'kernel_ms_per_launch': ms,
                        'kernel_ms_per_step': ms / K, 'achieved': sw * 2.0 * n * n / 1e12 / (ms / 1e3),
                        'frac': sw * 2.0 * n * n / 1e12 / (ms / 1e3) / FP64_PEAK_TFLOPS})
        # the same 512-restart steps one launch per step (tile-bound kernel, what rounds 1-3 would run on each of 8 GPUs)
        e.cd_queue(0)
        e.randn(512, seed=77)
        e.cd_run(phase1=True, seed=77)
        e.sync()
        t0 = time.perf_counter()
        reps = 6
        for k in range(reps):
            e.randn(512, seed=78 + k)
            e.cd_run(phase1=True, seed=78 + k)
            e.select_best(1e-4, want_x=False)
        e.sync()
        dt1 = (time.perf_counter() - t0) / reps
        recs.append({'config': 'headline workload through the lifecycle kernel: 40 steps of 4096 restarts in one launch (steady state), and the '
                               'strong-scaling share of BASELINE.json configs[1] on 8 GPUs -- 512 restarts per GPU and step, 160 steps in one launch',
                     'metric': 'restarts x coord-sweeps / s (phase 2)', 'value': pts[0]['value'], 'unit': 'restart-sweeps/s',
                     'kernel': stream_kernel, 'by_restarts_per_step': pts,
                     'one_launch_per_step_512_restarts_ms': 1e3 * dt1, 'one_launch_per_step_kernel': e.last_cd_kernel(),
                     'note': 'a streamed step of 512 restarts takes %.3f ms against %.3f ms for a step of 4096 (ratio %.2f; 8.0 = perfect strong '
                             'scaling of the step rate) and against %.2f ms with one launch per step (suggest + improve + selection, tile-bound '
                             'phase 2: 32 tiles on 256 CUs)' % (pts[1]['ms_per_step'], pts[0]['ms_per_step'], pts[0]['ms_per_step'] / pts[1]['ms_per_step'], 1e3 * dt1),
                     'roofline': {'bound': 'mfma', 'kernel': stream_kernel, 'achieved': pts[0]['achieved'], 'peak': FP64_PEAK_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': pts[0]['frac'], 'kernel_ms_per_launch': pts[0]['kernel_ms_per_launch']}})
        del e
    except Exception as ex:
        recs.append({'config': 'headline workload, streamed steps', 'error': repr(ex)})
    # configs[2]: MAXCUT n = 2000, 8192 Goemans-Williamson samples: x = F xi (MFMA GEMM sampler) + batched evaluation
    try:
        n, S, rk = 2000, 8192, 40
        funcs, _, _ = problems.maxcut(n, 0.5, seed=1)
        e = Engine(QCQPForm.from_arrays(funcs), device=device)
        rs = np.random.RandomState(5)
        V = rs.randn(n, rk)
        V /= np.linalg.norm(V, axis=1)[:, None]
        F = np.zeros((n, n))
        F[:, :rk] = V
        mu = np.zeros(n)
        e.sdr_sample(mu, F, S, seed=1, first_index=0)
        e.eval()
        e.sync()
        reps, t_s, t_e = 3, 0.0, 0.0
        t0 = time.perf_counter()
        for k in range(reps):
            e.sdr_sample(None, None, S, seed=2 + k, first_index=0)      # the factor of the relaxation is resident
            t_s += e.kernel_ms(Engine.KERNEL_SDR)
            e.eval()
            t_e += e.kernel_ms(Engine.KERNEL_EVAL)
        e.sync()
        dt = time.perf_counter() - t0
        fl = 2.0 * n * n * S
        recs.append({'config': 'BASELINE.json configs[2]: MAXCUT G(2000, 0.5), 8192 SDR samples drawn and evaluated on one GPU',
                     'metric': 'samples drawn + evaluated / s', 'value': reps * S / dt, 'unit': 'samples/s',
                     'sampler_kernel_ms': t_s / reps, 'eval_kernel_ms': t_e / reps,
                     'roofline': {'bound': 'mfma', 'kernel': 'dense_products_kernel (sampler x = mu + F xi; evaluation X^T P X)',
                                  'achieved': fl / 1e12 / (t_s / reps / 1e3), 'achieved_eval': fl / 1e12 / (t_e / reps / 1e3),
                                  'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': fl / 1e12 / (t_e / reps / 1e3) / FP64_PEAK_TFLOPS,
                                  'frac_sampler': fl / 1e12 / (t_s / reps / 1e3) / FP64_PEAK_TFLOPS,
                                  'algorithmic_flops_per_launch': fl,
                                  'note': 'frac = the EVALUATION product (2 n^2 S flops whatever the factor); frac_sampler counts the 2 n^2 S '
                                          'flops of the general n x n factor the API takes and the sampler multiplies (the eigen-factor of the '
                                          'relaxation; SURVEY 8d counts n^2 S for a triangular Cholesky factor: half of it)'}})
        del e
    except Exception as ex:      # a secondary record must never take the headline down
        recs.append({'config': 'configs[2]', 'error': repr(ex)})
    # configs[2]'s graph through improve(COORD_DESCENT) (examples/maxcut.py:25-28): n = 2000 > 1024 and a ZERO diagonal -- the second
    # generation lifecycle kernel (eight-wave workgroups, linear scalar objective); rounds 1-4 ran it through cd_phase2_rs_kernel
    try:
        n, R, K = 2000, 2048, 2
        pts = []
        for weighted in (True, False):
            funcs, _, _ = problems.maxcut(n, 0.5, seed=1, weighted=weighted)
            e = Engine(QCQPForm.from_arrays(funcs), device=device)
            e.cd_stream_run(1, 512, seed=5, num_iters=50)
            e.sync()
            t0 = time.perf_counter()
            o = e.cd_stream_run(K, R, seed=6, seed_stride=1, num_iters=50)
            e.sync()
            dt = time.perf_counter() - t0
            ms = e.kernel_ms(Engine.KERNEL_CD2)
            sw = float(o['visits2'].sum()) / n
            pts.append({'graph': 'weighted U(0.5, 1.5)' if weighted else 'unweighted (examples/maxcut.py)', 'kernel': e.last_cd_kernel(),
                        'value': sw / dt, 'kernel_ms_per_launch': ms, 'sweeps_per_restart': sw / (K * R),
                        'best_cut': float(-o['best_f0'].min()), 'achieved': sw * 2.0 * n * n / 1e12 / (ms / 1e3),
                        'frac': sw * 2.0 * n * n / 1e12 / (ms / 1e3) / FP64_PEAK_TFLOPS})
            del e
        recs.append({'config': 'MAXCUT G(2000, 0.5) (the graph of BASELINE.json configs[2]) through improve(COORD_DESCENT): 2 populations of 2048 random '
                               'restarts in one launch, num_iters = 50',
                     'metric': 'restarts x coord-sweeps / s (phase 2)', 'value': pts[0]['value'], 'unit': 'restart-sweeps/s',
                     'kernel': pts[0]['kernel'], 'by_graph': pts,
                     'note': 'unweighted graphs are full of exact ties (t1 = 0 up to the rounding of a row sum): those visits replay the '
                             'reference\'s arithmetic and its random tie-breaks one restart at a time -- the rate of the second entry is that replay',
                     'roofline': {'bound': 'mfma', 'kernel': pts[0]['kernel'], 'achieved': pts[0]['achieved'], 'peak': FP64_PEAK_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': pts[0]['frac'], 'kernel_ms_per_launch': pts[0]['kernel_ms_per_launch'],
                                  'algorithmic_flops_per_restart_sweep': 2.0 * n * n}})
    except Exception as ex:
        recs.append({'config': 'MAXCUT G(2000, 0.5) through improve(COORD_DESCENT)', 'error': repr(ex)})
    # separable problems with SEVERAL constraint classes / two constraints per coordinate through the lifecycle launch (round 6: kinds
    # GENK / LINK of cd_life_kernel; rounds 1-5 ran them one population at a time through the general phase-2 kernel), beside the serial path
    try:
        n, R, K = 1024, 2048, 2
        pts = []
        for fam, iters in (('ann2', 1000), ('box3', 40), ('cut2', 1000)):
            funcs = problems.multi_class(fam, n)
            e = Engine(QCQPForm.from_arrays(funcs), device=device)
            e.cd_stream_run(1, 512, seed=5, num_iters=iters)
            e.sync()
            t0 = time.perf_counter()
            o = e.cd_stream_run(K, R, seed=6, seed_stride=1, num_iters=iters)
            e.sync()
            dt = time.perf_counter() - t0
            ms = e.kernel_ms(Engine.KERNEL_CD2)
            kname = e.last_cd_kernel()
            sw = float(o['visits2'].sum()) / n
            # the same restarts one population at a time through qcqpmi_pop_randn + qcqpmi_cd_run (what the API did before)
            t0 = time.perf_counter()
            for p in range(K):
                e.randn(R, seed=6 + p)
                e.cd_run(phase1=True, num_iters=iters, seed=6 + p)
            e.sync()
            dts = time.perf_counter() - t0
            pts.append({'family': fam, 'kernel': kname, 'value': sw / dt, 'kernel_ms_per_launch': ms, 'sweeps_per_restart': sw / (K * R),
                        'num_iters': iters, 'achieved': sw * 2.0 * n * n / 1e12 / (ms / 1e3), 'frac': sw * 2.0 * n * n / 1e12 / (ms / 1e3) / FP64_PEAK_TFLOPS,
                        'serial_path_s': dts, 'serial_path_kernel': e.last_cd_kernel(), 'streamed_s': dt, 'speedup_over_serial_path': dts / dt})
            del e
        recs.append({'config': 'separable problems with several constraint classes / two constraints per coordinate (problems.multi_class: an annulus class '
                               'beside an equality class; three box classes; MAXCUT with a relaxed class), n = 1024, 2 populations of 2048 random restarts '
                               'in one launch of the lifecycle kernel',
                     'metric': 'restarts x coord-sweeps / s (phase 2)', 'value': pts[0]['value'], 'unit': 'restart-sweeps/s',
                     'kernel': pts[0]['kernel'], 'by_family': pts,
                     'roofline': {'bound': 'mfma', 'kernel': pts[0]['kernel'], 'achieved': pts[0]['achieved'], 'peak': FP64_PEAK_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': pts[0]['frac'], 'kernel_ms_per_launch': pts[0]['kernel_ms_per_launch'],
                                  'algorithmic_flops_per_restart_sweep': 2.0 * n * n}})
    except Exception as ex:
        recs.append({'config': 'several constraint classes through the lifecycle launch', 'error': repr(ex)})
    # the headline family at n = 4096 (four times configs[1]'s n, the same 256 rows of A): only the FACTORED instantiation of the lifecycle
    # kernel goes there (round 6) -- it keeps Y = L^T X (rank 256), not the X tile, in registers: its work per block does not grow with n
    try:
        n, R, K = 4096, 4096, 2          # 8192 restarts = 512 tiles: every tile slot of the chip busy
        funcs, _, _ = problems.boolean_least_squares(n, 256, seed=1)
        from qcqp_amd import lowrank
        P0s = funcs[0][0]
        Lf = lowrank.objective_factor(P0s.toarray() if hasattr(P0s, 'toarray') else np.asarray(P0s), max_rank=288)
        e = Engine(QCQPForm.from_arrays(funcs), device=device)
        e.cd_set_objective_factor(Lf)
        e.cd_stream_run(1, 512, seed=5)
        e.sync()
        t0 = time.perf_counter()
        o = e.cd_stream_run(K, R, seed=6, seed_stride=1)
        e.sync()
        dt = time.perf_counter() - t0
        ms = e.kernel_ms(Engine.KERNEL_CD2)
        sw = float(o['visits2'].sum()) / n
        rb = (int(Lf.shape[1]) + 15) // 16
        executed = (n / 16.0) * (8 * rb + 8) * 2048.0 / 16.0          # flops the kernel executes per restart-sweep (see the headline's note)
        recs.append({'config': 'headline family at n = 4096 (Boolean least squares, 256 rows of A: P0 = L L^T of rank 256), 2 populations of 4096 random '
                               'restarts in one launch of the FACTORED lifecycle kernel, to convergence',
                     'metric': 'restarts x coord-sweeps / s (phase 2)', 'value': sw / dt, 'unit': 'restart-sweeps/s', 'kernel': e.last_cd_kernel(),
                     'kernel_ms_per_launch': ms, 'sweeps_per_restart': sw / (K * R), 'feasible': int(o['ran_phase2'].sum()),
                     'roofline': {'bound': 'mfma', 'kernel': e.last_cd_kernel(), 'achieved': sw * executed / 1e12 / (ms / 1e3), 'peak': FP64_PEAK_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': sw * executed / 1e12 / (ms / 1e3) / FP64_PEAK_TFLOPS,
                                  'executed_flops_per_restart_sweep': executed, 'algorithmic_flops_per_restart_sweep': 2.0 * n * n,
                                  'algorithmic_over_peak': sw * 2.0 * n * n / 1e12 / (ms / 1e3) / FP64_PEAK_TFLOPS,
                                  'note': 'frac counts the EXECUTED matrix work here; algorithmic_over_peak is the same time priced by the algorithmic '
                                          '2 n^2 flops of a sweep (the headline\'s convention: SURVEY 8d) -- not a fraction of the hardware: the factored '
                                          'kernel\'s work per block of 16 coordinates is 8 r / 16 + 8 matrix instructions whatever n is'}})
        del e
    except Exception as ex:
        recs.append({'config': 'headline family at n = 4096 (factored lifecycle kernel)', 'error': repr(ex)})
    # configs[1]'s problem through improve(ADMM): separable constraints x_i^2 = 1 -> bases of unit vectors (round 5)
    try:
        n, R, iters = 1024, 4096, 100
        funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
        form = QCQPForm.from_arrays(funcs)
        e = Engine(form, device=device)
        t0 = time.perf_counter()
        lam, Bv, qhat = form.unit_bases()
        e.admm_set_basis(lam, Bv, qhat)
        rho = 50.0 / form.m                      # improve_admm's automatic rho for a positive semidefinite P0 (qcqp.py:270-277)
        e.admm_zsolver_device(rho)
        e.sync()
        t_setup = time.perf_counter() - t0
        pts = {}
        for unit in (True, False):
            e.admm_unit_bases(unit)
            e.randn(R, seed=3)
            e.admm_run(rho, None, phase1=True, num_iters=2)
            e.randn(R, seed=3)
            e.sync()
            t0 = time.perf_counter()
            out = e.admm_run(rho, None, phase1=True, num_iters=iters)
            e.sync()
            pts[unit] = (time.perf_counter() - t0, float(out['iters1'].sum()), float(out['iters2'].sum()), e.last_admm_kernel()[0], out)
        dt, i1, i2, name, out = pts[True]
        fl = 2.0 * n * n                          # per phase-2 restart-iteration: z = Minv rhs (qcqp.py:231-232); f0(z) for `better` comes out of the solve's right-hand side
        recs.append({'config': 'BASELINE.json configs[1]\'s problem (Boolean least squares n = 1024, m = 1024 constraints) through improve(ADMM, '
                               'num_iters=%d), %d restarts on one GPU' % (iters, R),
                     'metric': 'restart-iterations / s', 'value': (i1 + i2) / dt, 'unit': 'restart-iterations/s', 'kernel': name,
                     'wall_s': dt, 'gemm_path_wall_s': pts[False][0], 'gemm_path_kernel': pts[False][3],
                     'setup_s': t_setup, 'setup': 'bases of unit vectors written down (the reference: 1024 LAPACK decompositions of 1024 x 1024 matrices, '
                                                  '8.6 GB of eigenvectors), (2 (P0 + rho m I))^-1 by Newton-Schulz on the device',
                     'iterations_per_restart': [i1 / R, i2 / R], 'feasible': int((out['maxviol'] < 1e-2).sum()), 'restarts': R,
                     'roofline': {'bound': 'mfma', 'kernel': 'gemm_pk_kernel (z = Minv rhs) + admm_unit_step_kernel (z-update, gather, projection, scatter) + take-z / bookkeeping kernels',
                                  'achieved': i2 * fl / dt / 1e12, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': i2 * fl / dt / 1e12 / FP64_PEAK_TFLOPS,
                                  'algorithmic_flops_per_restart_iteration': fl,
                                  'note': 'wall clock of the whole improve_admm (phase 1: no matrix product at all with unit bases; phase 2: ONE n x n '
                                          'product per restart-iteration -- the reference also evaluates f0(z) = z^T P0 z + ... per iteration, here P0 z = rhs / 2 - rho m z '
                                          'from the solve itself); 4 launches per phase-2 iteration (solve, take-z, unit step, bookkeeping), 2 per phase-1 iteration: the replay of the reference\'s bisection steps in the unit step and the element-wise passes are most of the time'}})
        del e
    except Exception as ex:
        recs.append({'config': 'configs[1] through improve(ADMM)', 'error': repr(ex)[:300]})
    # configs[3]: secondary-user beamforming, 512 antennas (n = 1024 real), 16 + 64 constraints, improve(ADMM, rho = 1)
    try:
        funcs, _, _ = problems.beamforming(512, 16, 64, seed=1)
        form = QCQPForm.from_arrays(funcs)
        e = Engine(form, device=device)
        t0 = time.perf_counter()
        lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
        e.admm_set_basis(lam, Bv, qhat)
        t_setup = time.perf_counter() - t0
        R, iters = 1024, 1000        # the reference's default num_iters: the run ends by its own stop rules
        recs_admm = {}
        for fused in (True, False):
            e.admm_fused(fused)
            e.randn(R, seed=3)
            e.admm_run(1.0, None, phase1=True, num_iters=2)
            e.randn(R, seed=3)
            e.sync()
            t0 = time.perf_counter()
            out = e.admm_run(1.0, None, phase1=True, num_iters=iters)
            e.sync()
            dt = time.perf_counter() - t0
            its = float(out['iters1'].sum() + out['iters2'].sum())
            name, cw = e.last_admm_kernel()
            recs_admm[fused] = (out, dt, its, name, cw, e.kernel_ms(Engine.KERNEL_ADMM) if fused else None)
        out, dt, its, name, cw, kms = recs_admm[True]
        idx, fb, vb, _ = e.select_best(1e-4, want_x=False)
        rp = int(info['rp'])
        Mh = form.m * rp
        flops_it = 4.0 * form.n * Mh              # two products per restart-iteration: W^T z and W d, 2 n (m rp) each
        # cluster-iterations: a tile of 16 restarts iterates until its slowest restart stops (lock step inside the tile)
        tile_its = sum(float((out['iters1'][t:t + 16].max() if len(out['iters1'][t:t + 16]) else 0) + out['iters2'][t:t + 16].max())
                       for t in range(0, R, 16)) * 16.0
        recs.append({'config': 'BASELINE.json configs[3]: beamforming 512 antennas, m = 80 (16 SINR + 64 interference), improve(ADMM, rho=1, '
                               'num_iters=1000), 1024 restarts on one GPU',
                     'metric': 'restart-iterations / s', 'value': its / dt, 'unit': 'restart-iterations/s',
                     'kernel': name, 'workgroups_per_tile': cw, 'wall_s': dt, 'kernel_ms': kms,
                     'multi_launch_value': recs_admm[False][2] / recs_admm[False][1],
                     'setup_s': t_setup, 'setup': 'reduced bases (rank <= 2) by device products + Jacobi, no eigendecomposition',
                     'iterations_per_restart': its / R, 'iterations_longest_restart': [int(out['iters1'].max()), int(out['iters2'].max())],
                     'feasible': int((out['maxviol'] < 1e-2).sum()), 'restarts': R,
                     'best': {'restart': idx, 'objective': fb, 'max_violation': vb},
                     'roofline': {'bound': 'mfma', 'kernel': name, 'achieved': tile_its * flops_it / (kms / 1e3) / 1e12, 'peak': 78.6,
                                  'unit': 'TFLOP/s', 'frac': tile_its * flops_it / (kms / 1e3) / 1e12 / 78.6,
                                  'algorithmic_flops_per_restart_iteration': flops_it,
                                  'achieved_useful': its * flops_it / (kms / 1e3) / 1e12, 'frac_useful': its * flops_it / (kms / 1e3) / 1e12 / 78.6,
                                  'note': 'frac_useful counts the restart-iterations the reference would run (a restart stops by its own rule); frac the flops of the lock-step tile iterations (16 restarts per tile until the slowest stops) / kernel time (HIP '
                                          'events); the two products stream their A fragments from L2 (512 B per MFMA), the secular solves, the two '
                                          'cluster exchanges per iteration and the bookkeeping are latency, not flops: profiles/r03_admm_summary.md'}})
        # full eigenbasis (what the reference computes: any rank) at n = 512, m = 40: the multi-launch path
        try:
            funcs2, _, _ = problems.beamforming(256, 8, 32, seed=1)
            form2 = QCQPForm.from_arrays(funcs2)
            e2 = Engine(form2, device=device)
            lm = np.zeros((form2.m, form2.n)); Q = np.zeros((form2.m, form2.n, form2.n))
            for k, f in enumerate(form2.fs):
                lm[k], Q[k] = np.linalg.eigh(np.asarray(f.P))
            e2.admm_set_eig(lm, Q)
            R2, it2 = 256, 30
            e2.randn(R2, seed=3)
            e2.admm_run(1.0, None, phase1=True, num_iters=2)
            e2.randn(R2, seed=3)
            e2.sync()
            t0 = time.perf_counter()
            o2 = e2.admm_run(1.0, None, phase1=True, num_iters=it2)
            e2.sync()
            dt2 = time.perf_counter() - t0
            n2 = float(o2['iters1'].sum() + o2['iters2'].sum())
            fl2 = 4.0 * form2.n * form2.n * form2.m
            recs.append({'config': 'configs[3] family in the FULL eigenbasis (constraints of any rank): n = 512, m = 40, 256 restarts, 30 + 30 iterations',
                         'metric': 'restart-iterations / s', 'value': n2 / dt2, 'unit': 'restart-iterations/s', 'kernel': e2.last_admm_kernel()[0],
                         'roofline': {'bound': 'mfma', 'kernel': 'gemm_pk_kernel', 'achieved': n2 * fl2 / dt2 / 1e12, 'peak': 78.6, 'unit': 'TFLOP/s',
                                      'frac': n2 * fl2 / dt2 / 1e12 / 78.6, 'algorithmic_flops_per_restart_iteration': fl2,
                                      'note': 'wall clock of the whole run; 4 n^2 m flops per restart-iteration (Q^T z and Q d for every constraint)'}})
            del e2
        except Exception as ex:
            recs.append({'config': 'configs[3] full eigenbasis', 'error': repr(ex)})
        del e
    except Exception as ex:
        recs.append({'config': 'configs[3]', 'error': repr(ex)})
    # configs[4] family at 1/16 linear size: dense indefinite n = 1024, m = 256 generated on the device, CD on 512 restarts (the
    # share of one GPU of eight) and on 4096 (where the per-block latency of the chain is hidden by occupancy)
    try:
        n, m = 1024, 256
        form = problems.dense_indefinite_generated(n, m, seed=7)
        e = Engine(form, device=device)
        pts = []
        for R in (512, 4096):
            e.randn(R, seed=5)
            e.cd_run(phase1=True, num_iters=1, seed=5)
            e.randn(R, seed=6)
            e.sync()
            t0 = time.perf_counter()
            out = e.cd_run(phase1=True, num_iters=2, seed=6)
            e.sync()
            dt = time.perf_counter() - t0
            sw = float(out['sweeps1'].sum()) + float(out['visits2'].sum()) / n
            fl = sw * 2.0 * n * n * (m + 1)
            ph = (e.kernel_ms(Engine.KERNEL_CD1) + e.kernel_ms(Engine.KERNEL_CD2)) / 1e3     # HIP events around the two sweep loops
            pts.append({'restarts': R, 'value': sw / dt, 'achieved': fl / 1e12 / dt, 'frac': fl / 1e12 / dt / FP64_PEAK_TFLOPS,
                        'sweep_loops_s': ph, 'wall_s': dt, 'value_sweep_loops': sw / ph, 'frac_sweep_loops': fl / 1e12 / ph / FP64_PEAK_TFLOPS})
        recs.append({'config': 'BASELINE.json configs[4] family at n = 1024, m = 256 (full size is 137.6 GB of matrices): dense indefinite '
                               'QCQP generated on the device, COORD_DESCENT, 2 sweeps per phase; 512 restarts (first figures) and 4096',
                     'metric': 'restarts x coord-sweeps / s (phase 1 + phase 2)', 'value': pts[0]['value'], 'unit': 'restart-sweeps/s',
                     'by_restarts': pts, 'kernel': e.last_cd_kernel(),
                     'roofline': {'bound': 'mfma', 'kernel': 'dense_products_kernel (G_k = P_k X for all k) + ' + e.last_cd_kernel(),
                                  'achieved': pts[0]['achieved'], 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': pts[0]['frac'],
                                  'algorithmic_flops_per_restart_sweep': 2.0 * n * n * (m + 1),
                                  'timing': 'value / frac: wall clock of the whole cd_run -- the two sweep loops (products, chain, host loop over '
                                            'sweeps) AND the two evaluations of all m + 1 functions improve_coord_descent needs (start of '
                                            'phase 1; gate and slack, qcqp.py:189), each as expensive as the products of one sweep: with only '
                                            '2 sweeps per phase they are a third of the wall clock; *_sweep_loops: HIP events around the sweep '
                                            'loops alone'}})
        # the SDP relaxation in front of it (suggest(SDR), qcqp.py:72-97): the engine's own Burer-Monteiro / augmented-Lagrangian
        # solver on the same problem; --sdr-full adds the full-size solve (137.6 GB, about 100 s)
        recs.append(sdr_record(e, form, n, m))
        del e
    except Exception as ex:
        recs.append({'config': 'configs[4]', 'error': repr(ex)})
    # configs[4] at FULL size: n = 4096, m = 1024, 137.6 GB of matrices generated on the device, 512 restarts = the share of one
    # GPU of eight; evaluation of all 1025 functions + phase 1 + gate + phase 2 with 2 sweeps per phase (about 5 s with the generation)
    try:
        n, m, R = 4096, 1024, 512
        t0 = time.perf_counter()
        form = problems.dense_indefinite_generated(n, m, seed=7)
        e = Engine(form, device=device)
        e.sync()
        t_gen = time.perf_counter() - t0
        e.upload(0.1 * np.random.RandomState(0).randn(n, R))
        e.eval()
        ev_ms = e.kernel_ms(Engine.KERNEL_EVAL)
        t0 = time.perf_counter()
        out = e.cd_run(phase1=True, num_iters=2, seed=1)
        e.sync()
        dt = time.perf_counter() - t0
        fl = 2.0 * (m + 1) * n * n
        s1, s2 = float(out['sweeps1'].sum()), float(out['visits2'].sum()) / n
        ms1, ms2 = e.kernel_ms(Engine.KERNEL_CD1), e.kernel_ms(Engine.KERNEL_CD2)
        recs.append({'config': 'BASELINE.json configs[4] at FULL size: dense indefinite QCQP n = 4096, m = 1024 (137.6 GB of fp64 matrices generated '
                               'on the device in %.1f s), 512 restarts (the share of one GPU of eight), COORD_DESCENT with 2 sweeps per phase' % t_gen,
                     'metric': 'restarts x coord-sweeps / s (phase 1 + phase 2)', 'value': (s1 + s2) / dt, 'unit': 'restart-sweeps/s',
                     'wall_s': dt, 'kernel': e.last_cd_kernel(), 'feasible': int((out['maxviol'] < 1e-2).sum()), 'restarts': R,
                     'evaluation': {'kernel_ms': ev_ms, 'achieved': fl * R / 1e12 / (ev_ms / 1e3), 'frac': fl * R / 1e12 / (ev_ms / 1e3) / FP64_PEAK_TFLOPS,
                                    'note': 'all 1025 quadratic forms of 512 candidates (dense_products_kernel<1>); one pass over the matrices from HBM '
                                            'would take 17.2 ms'},
                     'roofline': {'bound': 'mfma', 'kernel': 'dense_products_kernel<0> + ' + e.last_cd_kernel(),
                                  'achieved': fl * s2 / 1e12 / (ms2 / 1e3) if ms2 > 0 else None, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': fl * s2 / 1e12 / (ms2 / 1e3) / FP64_PEAK_TFLOPS if ms2 > 0 else None,
                                  'frac_phase1': fl * s1 / 1e12 / (ms1 / 1e3) / FP64_PEAK_TFLOPS if ms1 > 0 else None,
                                  'algorithmic_flops_per_restart_sweep': fl,
                                  'timing': 'HIP events around the sweep loops of phase 2 (frac) and phase 1 (frac_phase1): block products on the '
                                            'matrix cores + chain kernels (sustained fp64 MFMA rate of the chip: profiles/r05_fp64_mfma_sustained.md)'}})
        del e
    except Exception as ex:
        recs.append({'config': 'configs[4] at full size', 'error': repr(ex)[:300]})
    if sdr_full:
        try:
            n, m = 4096, 1024
            form = problems.dense_indefinite_generated(n, m, seed=7)
            e = Engine(form, device=device)
            recs.append(sdr_record(e, form, n, m))
            del e
        except Exception as ex:
            recs.append({'config': 'configs[4] SDP relaxation at full size', 'error': repr(ex)})
    return recs


def sdr_record(e, form, n, m):
    """One record of the SDP relaxation (suggest(SDR), qcqp.py:72-97) of a dense problem held by engine `e`."""
    try:
        from qcqp_amd import sdr
        t0 = time.perf_counter()
        X, bound, info = sdr.solve_sdr_general(e, form, outer=40, inner=300)
        dt = time.perf_counter() - t0
        lmin, S = sdr.dual_certificate_device(e, info['y'], info['yN'])
        fl_eval = 2.0 * (m + 1) * n * n * info['rank']
        return {'config': 'SDP relaxation of the configs[4] family at n = %d, m = %d (%.1f GB of matrices): Burer-Monteiro factor of rank %d, '
                          'augmented Lagrangian, function values and gradient on the device, L-BFGS on the host'
                          % (n, m, (m + 1) * n * n * 8 / 1e9, info['rank']),
                'metric': 'seconds to a certified relaxation', 'value': dt, 'unit': 's', 'higher_is_better': False,
                'evaluations': info['evals'], 'ms_per_evaluation': 1e3 * dt / info['evals'], 'outer_iterations': len(info['hist']),
                'bound': bound, 'dual_value': info['dual_value'], 'infeasibility': info['infeas'],
                'lambda_min_of_dual_matrix': lmin, 'dual_matrix_scale': float(np.abs(S).max()),
                'ms_per_evaluation_by_part': {k: 1e3 * v / info['evals'] for k, v in info['timing'].items()},
                'roofline': {'bound': 'mfma', 'kernel': "dense_products_kernel<1> (all quadratic forms of the factor's columns)",
                             'achieved': fl_eval * info['evals'] / info['timing']['eval_parts'] / 1e12, 'peak': FP64_PEAK_TFLOPS,
                             'unit': 'TFLOP/s', 'frac': fl_eval * info['evals'] / info['timing']['eval_parts'] / 1e12 / FP64_PEAK_TFLOPS,
                             'timing': 'wall clock of the evaluation calls (upload excluded)'},
                'full_size': 'n = 4096, m = 1024 (137.6 GB): 95 s, 1582 evaluations of 60 ms (round 1: 700 s, 3481 of 201 ms), '
                             'profiles/r03_cfg5_sdr.md'}
    except Exception as ex:      # a secondary record must never take the headline down
        return {'config': 'configs[4] SDP relaxation (n = %d, m = %d)' % (n, m), 'error': repr(ex)}


def secondary_cpu_baselines(cores):
    """The reference's CPU path beside the SECONDARY records, through the oracle (C restatement in the
    reference's per-call structure, one core) and, where the survey asks for it, the hoisted-BLAS NumPy form on `cores`
    threads -- bounded samples of a few seconds each.  Returns {key: cpu_baseline} for the records that start with the key."""
    from oracle import oracle as orc
    from qcqp_amd import problems
    out = {}
    try:        # configs[2]: the reference evaluates sample by sample (qcqp.py:396-401: f0.eval + m one-coordinate evals)
        n, S = 2000, 8192
        funcs, _, _ = problems.maxcut(n, 0.5, seed=1)
        prob = orc.Problem(funcs)
        rs = np.random.RandomState(5)
        X = rs.randn(n, 3)
        t0 = time.perf_counter()
        prob.eval_batch(X)
        dt = (time.perf_counter() - t0) / 3
        P0 = np.asarray(funcs[0][0].todense() if hasattr(funcs[0][0], 'todense') else funcs[0][0])
        Xb = rs.randn(n, S)
        t0 = time.perf_counter()
        f = np.einsum('ij,ij->j', P0.dot(Xb), Xb)
        mv = np.max(np.abs(Xb * Xb - 1.0), axis=0)
        dtb = time.perf_counter() - t0
        out['BASELINE.json configs[2]'] = {
            'value': 1.0 / dt, 'unit': 'samples evaluated/s', 'cores': 1, 'kind': 'port',
            'sample': '3 samples through the oracle (objective + 2000 one-coordinate constraints per sample, the reference\'s per-sample '
                      'structure; its np.random.multivariate_normal draw -- an SVD of the 2000 x 2000 covariance PER SAMPLE, 2.9 s each, '
                      'SURVEY.md section 6 -- is not included)',
            'hoisted_blas': {'value': S / dtb, 'unit': 'samples evaluated/s', 'cores': cores, 'kind': 'port (NumPy: one GEMM P0 X for all samples)',
                             'sample': '8192 samples, %.2f s' % dtb, 'check': float(f[0] + mv[0])}}
    except Exception as ex:
        out['BASELINE.json configs[2]'] = {'error': repr(ex)[:300]}
    try:        # configs[3]: improve_admm restart-iterations / s / core (full eigenbasis like the reference; exact rank-2 eigenpairs)
        funcs, _, _ = problems.beamforming(512, 16, 64, seed=1)
        prob = orc.Problem(funcs)
        n, m = prob.n, prob.m
        rs = np.random.RandomState(0)
        lm = np.zeros((m, n))
        Q = np.zeros((m, n, n))
        for k in range(m):      # eigenpairs of a rank-2 matrix in O(n^2): range, 2 x 2 eigenproblem, null space by a complete QR
            Pk = np.asarray(funcs[k + 1][0])
            U, _ = np.linalg.qr(Pk.dot(rs.randn(n, 4)))
            w, V = np.linalg.eigh(U.T.dot(Pk).dot(U))
            keep = np.argsort(-np.abs(w))[:2]
            W = U.dot(V[:, keep])
            Qc, _ = np.linalg.qr(W, mode='complete')
            vals = np.concatenate([w[keep], np.zeros(n - 2)])
            vecs = np.concatenate([W, Qc[:, 2:]], axis=1)
            order = np.argsort(vals, kind='stable')
            lm[k], Q[k] = vals[order], vecs[:, order]
        prob._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))
        x0 = rs.randn(n)
        t0 = time.perf_counter()
        prob.improve_admm(x0, num_iters=3, rho=1.0)
        dt = time.perf_counter() - t0
        out['BASELINE.json configs[3]'] = {
            'value': 6.0 / dt, 'unit': 'restart-iterations/s', 'cores': 1, 'kind': 'port',
            'sample': 'one restart, 3 + 3 iterations of improve_admm through the oracle (80 onecons_qcqp calls of three dense 1024 x 1024 '
                      'products each per iteration, utilities.py:149-196), %.1f s; the 80 eigendecompositions the reference computes first '
                      '(about 1 s each) are not included' % dt}
    except Exception as ex:
        out['BASELINE.json configs[3]'] = {'error': repr(ex)[:300]}
    try:        # configs[4] family: get_onevar_func for every function at one coordinate = one coordinate update of the reference
        n, m = 1024, 32
        funcs, _, _ = problems.dense_indefinite(n, m, seed=7)
        prob = orc.Problem(funcs)
        x = 0.1 * np.random.RandomState(1).randn(n)
        t0 = time.perf_counter()
        cnt = 0
        for c in (3, 500, 1000):
            for k in range(m + 1):
                prob.onevar_coeffs(k, x, c)
                cnt += 1
        dt = time.perf_counter() - t0
        per_fc = dt / cnt
        out['BASELINE.json configs[4]'] = {
            'value': 1.0 / (per_fc * 257 * 1024), 'unit': 'restart-sweeps/s', 'cores': 1, 'kind': 'port',
            'per_function_and_coordinate_s': per_fc,
            'sample': 'get_onevar_func (utilities.py:99-105: a full P z per call) for %d (function, coordinate) pairs of a dense n = 1024 problem through '
                      'the oracle, %.2f s; a sweep at n = 1024, m = 256 is 257 x 1024 such calls (the interval arithmetic of onevar_qcqp is negligible '
                      'beside them): extrapolated' % (cnt, dt)}
    except Exception as ex:
        out['BASELINE.json configs[4]'] = {'error': repr(ex)[:300]}
    return out


# ------------------------------------------------------------------------------------- CPU baselines
def effective_cores():
    """Host cores this process may actually use: the affinity mask, cut down by the container's CPU quota
    (cgroup v2 cpu.max / v1 cfs_quota) -- nproc can say 256 where the quota allows 2."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _cpu_worker(job):
    """A bounded piece of the workload through the oracle (runs in a worker process, one per host core)."""
    kind, n, m_rows, seed, r = job
    from oracle import oracle as orc
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(n, m_rows, seed=1)
    prob = orc.Problem(funcs)
    rng = orc.Rng(orc.RNG_KEYED, seed)
    rng.set_restart(r)
    rs = np.random.RandomState(1000 + r)
    if kind == 'port':
        # reference-faithful per-call structure (one CSR per function, get_onevar_func per (constraint, coordinate)):
        # two phase-2 sweeps from the timed-metric start of BASELINE.md section 3 (feasible within the slack)
        x0 = np.sign(rs.randn(n)) * (1.0 + 2e-5 * rs.rand(n))
        t0 = time.time()
        x, s2 = prob.cd_phase2(x0, num_iters=2, rng=rng)
        return dict(r=r, dt=time.time() - t0, sweeps2=s2[1] / float(n))
    if kind == 'winner':
        # the bench's winning restart again on the CPU: faithful phase 1, then phase 2 with incremental bookkeeping
        # (same one-variable solver and trajectory as the faithful phase 2, tests/test_oracle_golden.py)
        x0 = orc.keyed_normal_matrix(seed, n, 1, first_index=r)[:, 0]
        t0 = time.time()
        x1, s1 = prob.cd_phase1(x0, rng=rng)
        ok = prob.max_violation(x1) < 1e-2
        x2, s2 = prob.cd_phase2_incremental(x1, rng=rng) if ok else (x1, np.zeros(3))
        return dict(r=r, dt=time.time() - t0, f0=prob.eval(0, x2), mv=prob.max_violation(x2), sweeps2=s2[1] / float(n))
    # optimised baseline: phase 2 to convergence, incremental gradient, same starts
    tot, dt = 0.0, 0.0
    for _ in range(8):
        x0 = np.sign(rs.randn(n)) * (1.0 + 2e-5 * rs.rand(n))
        t1 = time.time()
        x, s2 = prob.cd_phase2_incremental(x0, rng=rng)
        dt += time.time() - t1
        tot += s2[1] / float(n)
    return dict(r=r, dt=dt, sweeps2=tot)


def cpu_baseline(n, m_rows, seed, winner, cores):
    """The oracle on all usable host cores (one process per core over disjoint restarts), bounded samples of the
    same workload: the faithful port (2 phase-2 sweeps per core, ~6 s), the optimised incremental-gradient C
    baseline (8 phase-2 runs per core), and the bench's winning restart re-run for the cross-check of `best`."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')      # the parent holds a HIP context: do not fork it
    os.environ['OMP_NUM_THREADS'] = '1'
    with ctx.Pool(processes=cores) as pool:
        pool.map(_cpu_worker, [('opt', n, m_rows, seed, r) for r in range(cores)], chunksize=1)   # warm: imports, page-in
        t0 = time.time()
        port = pool.map(_cpu_worker, [('port', n, m_rows, seed, r) for r in range(cores)], chunksize=1)
        wall_port = time.time() - t0
        t0 = time.time()
        opt = pool.map(_cpu_worker, [('opt', n, m_rows, seed, r) for r in range(cores)], chunksize=1)
        wall_opt = time.time() - t0
        win = pool.map(_cpu_worker, [('winner', n, m_rows, seed, winner)], chunksize=1)[0]
    return port, wall_port, opt, wall_opt, win


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)      # 0.5 s of timed work at N = 1 (all steps resident: 64 MiB per step)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n', type=int, default=1024)
    ap.add_argument('--m-rows', type=int, default=256)
    ap.add_argument('--restarts', type=int, default=4096, help='restarts per GPU (weak) / in total (strong)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--cpu-cores', type=int, default=0, help='host cores for the CPU baselines (0 = all usable, at most 32)')
    ap.add_argument('--no-overlap', dest='overlap', action='store_false',
                    help='scheme two with one context only: every step runs strictly after the previous one')
    ap.add_argument('--scheme', choices=['auto', 'stream', 'two'], default='auto',
                    help='stream: the K steps as ONE persistent launch of the lifecycle kernel (qcqpmi_cd_stream_run: every restart runs its '
                         'whole step -- suggest, phase 1, gate, phase 2, objective -- in a slot of a workgroup; DESIGN.md section 4.1d); two: '
                         'round 2/3 scheme, two contexts per GPU, one phase-2 launch per step, the preparation of step k + 1 in the tail of launch k; '
                         'auto (default): stream, and on one GPU the same steps through `two` as well (in the line under `schemes`; the best point '
                         'must be the same)')
    ap.add_argument('--no-factor', action='store_true',
                    help='do not hand the factor of the objective (P0 = L L^T, rank = rows of A) to the lifecycle kernel: it then '
                         'multiplies with P0 itself (rounds 4 / 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--comm', choices=['rccl', 'file'], default='rccl',
                    help='transport of the best-point exchange: RCCL (default), or the job\'s file rendezvous -- for ranks that share '
                         'one device (RCCL refuses duplicate devices; tests: two real engines on the one GPU of a test box)')
    ap.add_argument('--device', type=int, default=-1, help='HIP device of EVERY rank (default: LOCAL_RANK)')
    ap.add_argument('--best-out', default='', help='rank 0 saves the global best point of every timed step (K x n, .npy)')
    ap.add_argument('--secondary', action='store_true',
                    help='also measure the bounded records of the other BASELINE.json configs (about 2.5 minutes): written to '
                         '--secondary-out as a sidecar JSON file, never into the headline line')
    ap.add_argument('--secondary-out', default=os.path.join(REPO, 'gpurun_out', 'bench_secondary.json'))
    ap.add_argument('--compare', action='store_true',
                    help='run the same K steps through scheme `two` and through the other lifecycle kernel as well (sidecar file)')
    ap.add_argument('--no-secondary', action='store_true', help='(default since round 6; accepted for old command lines)')
    ap.add_argument('--sdr-full', action='store_true', help='add the SDP relaxation of configs[4] at FULL size (137.6 GB, about 100 s) to the secondary records')
    args = ap.parse_args()
    line_fd = claim_stdout()

    from qcqp_amd import dist, problems
    from qcqp_amd.engine import E_UNSUPPORTED, Engine, EngineError
    from qcqp_amd.form import QCQPForm

    kids = dist.spawn_local_ranks(args.gpus)       # no-op under a launcher or for 1 GPU
    rank, local_rank, world = dist.env_world()
    n = args.n
    if args.scaling == 'weak':
        first, R = rank * args.restarts, args.restarts
    else:
        first, R = dist.shard_range(args.restarts, rank, world)

    funcs, _, _ = problems.boolean_least_squares(n, args.m_rows, seed=1)
    form = QCQPForm.from_arrays(funcs)
    if args.device >= 0:
        local_rank = args.device
    if args.comm == 'file' and args.scheme == 'auto':
        args.scheme = 'stream'                     # (scheme `two` selects through RCCL inside the library)
    eng = Engine(form, device=local_rank)
    boot = dist.init_file_comm(eng, rank, world) if args.comm == 'file' else dist.init_rccl(eng, rank, world)
    # the objective of a least-squares problem has rank rows(A): found from P0 alone (pivoted Cholesky, verified entry by entry;
    # problem set-up, outside the timed region) -- the lifecycle kernel then carries L^T X instead of multiplying with P0
    factor_rank = 0
    if not args.no_factor and hasattr(eng, 'cd_set_objective_factor'):
        from qcqp_amd import lowrank
        P0 = funcs[0][0]
        P0 = P0.toarray() if hasattr(P0, 'toarray') else np.asarray(P0)
        Lf = lowrank.objective_factor(P0, max_rank=min(288, n // 2)) if n >= 128 else None
        if Lf is not None:
            try:
                eng.cd_set_objective_factor(Lf)
                factor_rank = int(Lf.shape[1])
            except EngineError as ex:
                if ex.code != E_UNSUPPORTED:
                    raise
    K = max(args.steps, 1)

    def allreduce_sum(a):
        return eng.comm_allreduce(a, 'sum')

    parts_ms = {}        # wall clock of the last run_stream call by part (host side)

    # ------------------------------------------------------------------ scheme `stream` (default): one launch for all steps
    def run_stream(count, base):
        """`count` steps -- step k = suggest(RANDOM) with seed + k, improve(COORD_DESCENT), best point -- through ONE launch of the
        lifecycle kernel on this rank's restarts, then ONE exchange over the ranks for the global best of every step (an all-gather
        of the 32-byte keys, an all-reduce of the winners' points)."""
        ta = time.perf_counter()
        o = eng.cd_stream_run(count, R, generate=True, phase1=True, num_iters=1000, viol_tol=1e-2, tol=1e-4, seed=args.seed + base,
                              seed_stride=1, first_index=first, first_stride=0, select_tol=1e-4)
        tb = time.perf_counter()
        ms = eng.kernel_ms(Engine.KERNEL_CD2)
        keys, X = dist.global_best_of_populations(allreduce_sum, rank, world, o['best_f0'], o['best_maxviol'], first + o['best_index'], o['best_x'],
                                                     allgather=getattr(eng, 'comm_allgather', None))
        parts_ms.clear()
        parts_ms.update(launch_and_fetch=1e3 * (tb - ta), exchange=1e3 * (time.perf_counter() - tb))
        return o, keys, X, ms

    # ------------------------------------------------------------------ scheme `two` (rounds 2 / 3): one phase-2 launch per step
    # A second context (= a second HIP stream) on the same GPU.  The steps alternate between the two: while the phase-2 kernel of
    # step k runs -- its tail leaves most CUs idle: 4096 restarts are one tile per CU and the launch lasts as long as the slowest
    # restart -- suggest, phase 1, evaluation and gate of step k + 1 are already being worked on in the other stream.  The phase-2
    # kernels themselves never overlap (the next one is launched after the results of the current one have been fetched).
    def make_two():
        engs = [Engine(form, device=local_rank)]
        dist.init_rccl(engs[0], rank, world, bootstrap=boot)
        if args.overlap:
            try:
                e2 = Engine(form, device=local_rank)
                dist.init_rccl(e2, rank, world, bootstrap=boot)   # its own communicator (same rendezvous object)
                ok = 1.0
            except Exception as ex:     # never lose the run over the optimisation: fall back to strictly serial steps
                sys.stderr.write('bench: second context unavailable (%r): steps will not overlap\n' % (ex,))
                e2, ok = None, 0.0
            if world > 1 and float(eng.comm_allreduce([ok], 'sum')[0]) < world:     # every rank must take the same path
                e2 = None
            if e2 is not None:
                engs.append(e2)
        return engs

    def run_two(engs, count, base, record):
        def prepare(e, k):
            e.randn(R, seed=args.seed + k, first_index=first)
            e.cd_begin(phase1=True, seed=args.seed + k, first_index=first)
        if count <= 0:
            return
        prepare(engs[0], base)
        engs[0].cd_phase2()
        for k in range(count):
            cur = engs[k % len(engs)]
            more = k + 1 < count
            if len(engs) > 1:
                nxt = engs[(k + 1) % 2]
                if more:
                    prepare(nxt, base + k + 1)          # runs in the tail of the phase-2 kernel in flight
                out = cur.cd_fetch()
                if more:
                    nxt.cd_phase2()                     # launched before the selection of step k is even looked at
                b = cur.comm_select_best(1e-4, index_offset=first)
            else:
                out = cur.cd_fetch()
                b = cur.comm_select_best(1e-4, index_offset=first)
                if more:
                    prepare(cur, base + k + 1)
                    cur.cd_phase2()
            record(k, cur, out, b)

    def measure_two():
        engs = make_two()
        run_two(engs, args.warmup, -1000, lambda *a_: None)
        for e_ in engs:
            e_.sync()
        eng.comm_barrier()
        acc = dict(sweeps1=0.0, sweeps2=0.0, p2_flops=0.0, p2_ms=0.0, p1_ms=0.0, best=None, best_step=-1)

        def record(k, cur, out, b):
            acc['sweeps1'] += float(out['sweeps1'].sum())
            acc['sweeps2'] += float(out['visits2'].sum()) / n
            acc['p2_flops'] += float(out['visits2'].sum()) * 2.0 * n   # algorithmic: 2n flops per visit
            acc['p2_ms'] += cur.kernel_ms(Engine.KERNEL_CD2)
            acc['p1_ms'] += cur.kernel_ms(Engine.KERNEL_CD1)
            best = acc['best']
            if best is None or dist.better_key(b[1], b[2], b[0]) < dist.better_key(best[1], best[2], best[0]):
                acc['best'], acc['best_step'] = b, k
        t0 = time.perf_counter()
        run_two(engs, args.steps, 0, record)
        for e_ in engs:
            e_.sync()
        eng.comm_barrier()
        acc['dt'] = time.perf_counter() - t0
        acc['kernel'] = engs[0].last_cd_kernel()
        acc['contexts'] = len(engs)
        return acc

    scheme = args.scheme
    if scheme in ('auto', 'stream'):
        try:
            run_stream(max(args.warmup, 1), -1000)       # (--warmup 0: one probing step, outside the timed region)
            ok = 1.0
        except EngineError as ex:
            # only "this problem family is not the lifecycle kernels'" (several constraint classes, coupled constraints, ...) selects
            # the other scheme; any other failure (a HIP error, bad arguments) is a failure of the benchmark
            if scheme == 'stream' or ex.code != E_UNSUPPORTED:
                raise
            sys.stderr.write('bench: scheme stream not available (%s): scheme two\n' % (ex,))
            ok = 0.0
        if world > 1:                   # every rank must take the same path
            ok = 1.0 if float(eng.comm_allreduce([ok], 'sum')[0]) >= world else 0.0
        scheme = 'stream' if ok else 'two'

    launch_ms = None
    if scheme == 'stream':
        eng.cd_stream_reserve(K, R)      # buffers of the K timed steps (the warm-up ran W steps): no allocation inside the timed region
        eng.sync()
        eng.comm_barrier()
        t0 = time.perf_counter()
        o, keys, Xbest, launch_ms = run_stream(args.steps, 0)
        eng.sync()
        eng.comm_barrier()
        dt = time.perf_counter() - t0
        sweeps1, sweeps2 = float(o['sweeps1'].sum()), float(o['visits2'].sum()) / n
        p2_flops, p2_ms, p1_ms = float(o['visits2'].sum()) * 2.0 * n, launch_ms, 0.0
        best_step = min(range(K), key=lambda k: dist.better_key(keys[k][1], keys[k][2], keys[k][0]) + (k,))
        best = (keys[best_step][0], keys[best_step][1], keys[best_step][2], Xbest[best_step])
        kernel_name, contexts = eng.last_cd_kernel(), 1
    else:
        acc = measure_two()
        dt, sweeps1, sweeps2, p2_flops, p2_ms, p1_ms = acc['dt'], acc['sweeps1'], acc['sweeps2'], acc['p2_flops'], acc['p2_ms'], acc['p1_ms']
        best, best_step, kernel_name, contexts = acc['best'], acc['best_step'], acc['kernel'], acc['contexts']
    dt = float(eng.comm_allreduce([dt], 'max')[0])
    tot = eng.comm_allreduce([sweeps1, sweeps2, p2_flops, p2_ms], 'sum')
    p2_ms_max = float(eng.comm_allreduce([p2_ms], 'max')[0])
    sweeps1_all, sweeps2_all, flops_all = float(tot[0]), float(tot[1]), float(tot[2])

    rc = 0
    if rank == 0:
        achieved = (p2_flops / 1e12) / (p2_ms / 1e3) if p2_ms > 0 else 0.0      # rank 0's GPU, its own launches (HIP events)
        pmc = profiled_counters(kernel_name)
        side = {}                       # everything that is not the headline: the sidecar file
        # ---- the headline record: short strings only (DESIGN.md section 6 has the prose)
        res = {
            'metric': 'restarts x coord-sweeps / sec (improve COORD_DESCENT phase-2 sweeps, 2n^2 flops each; whole step timed)',
            'value': sweeps2_all / dt,
            'unit': 'restart-sweeps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / K,
            'timed_region_s': dt,
            'timed_region_parts_ms': {k: round(v, 3) for k, v in parts_ms.items()} if scheme == 'stream' else None,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': 'Boolean least squares n=%d m=%d, %d random restarts %s, COORD_DESCENT '
                                   '(BASELINE.json configs[1])' % (n, args.m_rows, args.restarts,
                                                                    'per GPU' if args.scaling == 'weak' else 'in total'),
                       'restarts_per_gpu': R, 'num_iters': 1000, 'viol_tol': 1e-2, 'tol': 1e-4,
                       'sharding': 'restarts by global index, replicas of P',
                       'scheme': scheme, 'objective_factor_rank': factor_rank,
                       'step': 'suggest(RANDOM) + phase 1 + gate + phase 2 to convergence + best-point selection'
                               + (', all K steps in ONE persistent launch per GPU' if scheme == 'stream' else '')},
            'phase2_sweeps_per_restart': sweeps2_all / (K * world * max(R, 1)),
            'phase1': {'restart_sweeps_per_s_incl': (sweeps1_all + sweeps2_all) / dt,
                       'sweeps_per_restart': sweeps1_all / (K * world * max(R, 1)),
                       'note': 'element-wise for separable constraints: not counted in value'},
            'best': {'objective': best[1], 'max_violation': best[2], 'global_restart_index': best[0],
                     'step': best_step},
            'roofline': {'bound': 'mfma', 'kernel': kernel_name or 'cd_general_kernel', 'achieved': achieved,
                         'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_PEAK_TFLOPS,
                         'traffic': pmc['traffic'] if pmc else None,
                         'traffic_source': pmc['source'] if pmc else None,
                         'mfma_busy': pmc['mfma_busy'] if pmc else None,
                         'algorithmic_flops_per_restart_sweep': 2.0 * n * n,
                         'algorithmic_flops_per_launch': p2_flops if scheme == 'stream' else p2_flops / K,
                         'kernel_ms_per_launch': p2_ms if scheme == 'stream' else p2_ms / K,
                         'launches': 1 if scheme == 'stream' else K,
                         'timing': 'HIP events on the engine stream around ' +
                                   ('the ONE launch that runs all %d timed steps' % K if scheme == 'stream' else 'every phase-2 launch')},
        }
        if factor_rank and 'factored' in (kernel_name or ''):
            # what the factored kernel EXECUTES per restart-sweep: per block of 16 coordinates of a tile of 16 restarts 4 RB MFMAs for
            # the product, 4 RB for the update of Y, 8 for the chain's fix-up (RB = blocks of 16 rows of Y), 2048 flops each
            rb = (factor_rank + 15) // 16
            executed = (n / 16.0) * (8 * rb + 8) * 2048.0 / 16.0
            res['roofline']['executed_flops_per_restart_sweep'] = executed
            res['roofline']['frac_executed'] = res['roofline']['frac'] * executed / (2.0 * n * n)
            res['roofline']['note'] = ('frac counts the ALGORITHMIC 2 n^2 flops of a sweep (the unit of `value`); the kernel carries L^T X '
                                       '(P0 = L L^T, rank %d) and executes %.2f of them; bound by the two sequential chains of a CU together with the three multiplying SIMDs'
                                       % (factor_rank, executed / (2.0 * n * n)))
        if pmc:
            side['traffic_provenance'] = {k: pmc[k] for k in ('source', 'profile_commit', 'profile_date', 'kernel', 'launch')}
        if world > 1:
            res['roofline']['kernel_ms_per_launch_max_over_ranks'] = p2_ms_max if scheme == 'stream' else p2_ms_max / K
            res['roofline']['achieved_all_gpus'] = (flops_all / 1e12) / (p2_ms_max / 1e3) if p2_ms_max > 0 else 0.0
        if world == 1 and not args.no_cpu_baseline:      # a reported baseline of rank 0 at N = 1 only
            cores = args.cpu_cores or min(effective_cores(), 32)
            # the winning restart of the winning step again on the CPU: the cross-check of `best`
            wseed = args.seed + best_step
            port, wall_port, opt, wall_opt, win = cpu_baseline(n, args.m_rows, wseed, int(best[0]), cores)
            sw_port = sum(p['sweeps2'] for p in port)
            res['best']['oracle_objective'] = win['f0']
            res['best']['oracle_max_violation'] = win['mv']
            res['best']['oracle_rel_err'] = abs(win['f0'] - best[1]) / (1.0 + abs(win['f0']))
            res['best']['oracle_viol_err'] = abs(win['mv'] - best[2])
            res['cpu_baseline'] = {
                'value': sw_port / wall_port, 'unit': 'restart-sweeps/s', 'cores': cores, 'kind': 'port',
                'per_core': sum(p['sweeps2'] / p['dt'] for p in port) / len(port),
                'sample': '2 phase-2 sweeps per core through oracle/ (C restatement, reference call structure), one process per '
                          'core: %.1f sweeps in %.1f s wall' % (sw_port, wall_port),
                'host_cpu_count': os.cpu_count(),
                'optimised': {'value': sum(o['sweeps2'] for o in opt) / wall_opt, 'cores': cores,
                              'kind': 'port with incremental gradient, O(n) per accepted move',
                              'per_core': sum(o['sweeps2'] / o['dt'] for o in opt) / len(opt)},
                # BASELINE.md section 3 (tools/calibrate_baseline.py, build container): the true reference's loop body is
                # 35.6x slower than this C port on identical coordinates (0.0124 vs 0.441 restart-sweeps/s/core)
                'true_reference': {'port_over_reference_speed': 35.6,
                                   'estimated_value': sw_port / wall_port / 35.6,
                                   'source': 'BASELINE.md section 3'},
            }
        # ---- the line is final here; what follows (opt-in, minutes) only fills the sidecar file and cannot change or lose it
        line = headline_line(res)
        if world == 1 and (args.compare or args.secondary):
            side['headline'] = json.loads(line)
            write_sidecar(args.secondary_out, side)
        if world == 1 and args.compare and args.scheme == 'auto' and scheme == 'stream':
            # the same K steps (same seeds) through the scheme of rounds 2 / 3, for comparison: the best point must be the same one
            try:
                acc = measure_two()
                ach2 = (acc['p2_flops'] / 1e12) / (acc['p2_ms'] / 1e3) if acc['p2_ms'] > 0 else 0.0
                b2 = acc['best']
                same_best = (int(b2[0]) == int(best[0]) and acc['best_step'] == best_step
                             and abs(b2[1] - best[1]) <= 1e-9 * (1.0 + abs(best[1])) and float(np.max(np.abs(np.asarray(b2[3]) - np.asarray(best[3])))) <= 1e-9)
                side['schemes'] = {'reported': 'stream',
                                   'two': {'value': acc['sweeps2'] / acc['dt'], 'ms_per_step': 1e3 * acc['dt'] / K, 'timed_region_s': acc['dt'],
                                           'roofline': {'kernel': acc['kernel'], 'achieved': ach2, 'frac': ach2 / FP64_PEAK_TFLOPS,
                                                        'kernel_ms_per_launch': acc['p2_ms'] / K,
                                                        'timing': 'HIP events around every phase-2 launch (the launches own the chip; preparation and '
                                                                  'selection are outside them)'},
                                           'best': {'objective': b2[1], 'max_violation': b2[2], 'global_restart_index': b2[0], 'step': acc['best_step']},
                                           'same_best_point_as_stream': bool(same_best), 'contexts': acc['contexts']}}
            except Exception as ex:
                side['schemes'] = {'reported': 'stream', 'two': {'error': repr(ex)[:400]}}
        if world == 1 and args.compare and scheme == 'stream' and hasattr(eng, 'cd_life_version'):
            # the same K steps (same seeds) through the OTHER lifecycle kernel (qcqpmi_cd_life_version: a debug switch)
            try:
                other = 2 if not (kernel_name or '').startswith('cd_life_kernel') else 1
                eng.cd_life_version(other)
                run_stream(max(args.warmup, 1), -1000)
                eng.sync()
                t0o = time.perf_counter()
                oo, keys_o, X_o, ms_o = run_stream(args.steps, 0)
                eng.sync()
                dt_o = time.perf_counter() - t0o
                sw_o = float(oo['visits2'].sum()) / n
                ach_o = sw_o * 2.0 * n * n / 1e12 / (ms_o / 1e3)
                bs_o = min(range(K), key=lambda k: dist.better_key(keys_o[k][1], keys_o[k][2], keys_o[k][0]) + (k,))
                side.setdefault('schemes', {'reported': 'stream'})['other_lifecycle_kernel'] = {
                    'kernel': eng.last_cd_kernel(), 'value': sw_o / dt_o, 'ms_per_step': 1e3 * dt_o / K, 'kernel_ms_per_launch': ms_o,
                    'roofline_frac': ach_o / FP64_PEAK_TFLOPS,
                    'same_best_point': bool(bs_o == best_step and int(keys_o[bs_o][0]) == int(best[0]) and
                                            float(np.max(np.abs(np.asarray(X_o[bs_o]) - np.asarray(best[3])))) <= 1e-9)}
            except Exception as ex:
                side.setdefault('schemes', {'reported': 'stream'})['other_lifecycle_kernel'] = {'error': repr(ex)[:300]}
            finally:
                eng.cd_life_version(0)
        if world == 1 and args.secondary:
            try:
                side['secondary'] = secondary_records(local_rank, sdr_full=args.sdr_full)
            except Exception as ex:
                side['secondary'] = [{'error': repr(ex)[:400]}]
            write_sidecar(args.secondary_out, side)
            if not args.no_cpu_baseline:
                try:
                    extra = secondary_cpu_baselines(args.cpu_cores or min(effective_cores(), 32))
                    for rec in side['secondary']:
                        for key, val in extra.items():
                            if str(rec.get('config', '')).startswith(key):
                                rec['cpu_baseline'] = val
                except Exception as ex:
                    sys.stderr.write('bench: CPU baselines of the secondary records failed: %r\n' % (ex,))
        if side and world == 1 and (args.compare or args.secondary):
            if write_sidecar(args.secondary_out, side):
                sys.stderr.write('bench: secondary records and comparisons: %s\n' % (args.secondary_out,))
        if args.best_out and scheme == 'stream':
            np.save(args.best_out, np.asarray(Xbest))
        emit_line(line_fd, line)
    if boot is not None:
        boot.close()
    rc = dist.wait_children(kids) if kids else 0
    return rc


if __name__ == '__main__':
    sys.exit(main())
