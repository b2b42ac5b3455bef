#!/usr/bin/env python3
"""Throughput bench of the Quadtree-MPNNLSTM training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full training step (forward + masked MSE + backward + [all-reduce] + clip_grad_norm_(10)
+ Adam) over one per-GPU batch of synthetic Moving-MNIST-like clips that is already resident in HBM.
Workload at every N: BASELINE.json configs[1] per GPU (64x64, 2 digits, in=10/out=10, 32 clips per GPU,
pixel noise 0.05, thresh 0.1, hidden 16, 2 layers, ChebConv K=3, dropout 0.1) -> weak scaling.
Rank 0 prints ONE JSON line.  `roofline` is measured on the message-aggregate kernel (k_spmm) with HIP
events around every launch of one extra, untimed training step; `cpu_baseline` times the CPU oracle
(a port of the reference algorithm, one clip per optimizer step like the reference) on the host cores.

Arithmetic: fp32 everywhere (fp32 MFMA for every GEMM, forward and backward) -- `dtype` "f32" means that.  The opt-in
split-bf16 data gradient of the gate GEMM (ops.DGRAD_SPLIT_BF16) is NOT part of `value`; the same captured step is timed
with it as well and reported beside the headline as `split_bf16_dgrad` (frozen model, same invocation).
Order: frozen phase -> warm-up -> the timed window (`value`) -> `--repeats` further windows (median / spread) -> then, on
rank 0 alone while the other ranks wait at a host-side store barrier (not inside an RCCL collective): the kernel probes on
a freshly initialised model, the split-bf16 variant, the CPU baseline.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'quadtree-mpnnlstm_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

T_IN, T_OUT, CANVAS, N_DIGITS, NOISE, THRESH = 10, 10, (64, 64), 2, 0.05, 0.1
HIDDEN, N_LAYERS, DROPOUT, LR = 16, 2, 0.1, 0.01
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU')
    ap.add_argument('--noise', type=float, default=NOISE)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-clips', type=int, default=10)
    ap.add_argument('--frozen-steps', type=int, default=10,
                    help='steps timed with the learning rate at 0 before training starts (frozen_ms_per_step; 0 = skip)')
    ap.add_argument('--repeats', type=int, default=4,
                    help='further timed windows of --steps steps after the one `value` comes from (median / spread; 0 = none)')
    ap.add_argument('--no-split-variant', action='store_true', help='skip the split-bf16 data-gradient comparison')
    ap.add_argument('--eager', action='store_true', help='Python-driven launches instead of hipGraph replay')
    ap.add_argument('--host-inputs', action='store_true',
                    help='keep the batches in pinned host memory and copy them in every step (PCIe-inclusive rate; informational)')
    ap.add_argument('--force-multi', action='store_true',
                    help='--gpus 1 only: a process group of ONE rank over RCCL and the multi-rank step structure (graph 1, flat '
                         'all-reduce, graph 2) -- what a rank of an N-GPU run does per step, measured on one GPU (informational)')
    return ap.parse_args()


def make_predictor(device, capturable=False):
    import torch
    from model.mpnnlstm import NextFramePredictorS2S
    torch.manual_seed(1)
    nfp = NextFramePredictorS2S(thresh=THRESH, input_features=1, input_timesteps=T_IN, output_timesteps=T_OUT,
                                device=device, model_kwargs=dict(hidden_size=HIDDEN, dropout=DROPOUT, n_layers=N_LAYERS))
    # QT_BENCH_LR=0 freezes the model (diagnostics: the meshes of the decoder then stay the same over a run, which makes
    # A/B comparisons of kernels independent of the training trajectory); the bench itself always trains at LR
    nfp.initiate_training(lr=float(os.environ.get('QT_BENCH_LR', LR)), lr_decay=0.95, capturable=capturable)
    return nfp


def spmm_roofline(nfp, batch, mask, reps=10, traffic_files=('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_spmm.json')):
    """Roofline of the message-aggregate kernels, measured live with HIP events on the launch stream.

    One extra (untimed) eager forward + backward records every message-aggregate launch of the real workload: the clip-resident
    recurrence launches (k_cheb_clip: ALL K - 1 hops of a ChebConv pass in one launch, neighbour rows in LDS; csrc/chebclip.hip)
    and any remaining per-hop k_spmm launches.  Each distinct launch is then re-issued `reps` times back to back (captured into a
    hipGraph, so that the device and not the host call is timed) between two events on the replay stream with same-shaped
    operands, and the per-launch averages are summed: the figure covers exactly the launch mix of one training step.

    Algorithmic bytes.  A per-hop launch moves SURVEY.md 8(d)'s 4(N+1) + 8E' + 8NC bytes (index arrays +
    every feature row read once + every output row written once).  A fused launch (all K - 1 hops) is priced on what IT must
    move through HBM -- index arrays once, Z once and K - 1 planes out (forward: 4(N+1) + 8E' + 4NC K), or K gradient planes in
    and one out (backward: 4(N+1) + 8E' + 4NC (K + 1)); the intermediate planes of a fused recurrence are never re-read from
    HBM.  `achieved` / `frac` use these per-launch bytes; `equivalent_unfused` keeps the hops x per-hop pricing of rounds 1 - 3
    for comparison; `traffic` (PMC) is what actually crossed the fabric.  Works for any workload / frame size (tools/
    bench_configs.py calls it for the 128x128 and 256x256 configurations, whose hops are per-hop k_spmm launches or tile-
    resident launches)."""
    import torch
    from qtmpnn import mesh as qmesh, ops
    records = []
    orig_spmm, orig_fwd, orig_bwd = qmesh.spmm2, ops.clip_planes, ops.clip_clenshaw

    def spy_spmm(ms, xs, alpha, ps, beta, qs, gamma, outs):
        records.append(('hop', ms, tuple(x.shape[1] for x in xs), ps is not None, qs is not None, 2))
        orig_spmm(ms, xs, alpha, ps, beta, qs, gamma, outs)

    def spy_fwd(ms, Zs, TZs, K):
        records.append(('fwd', ms, tuple(z.shape[1] for z in Zs), False, False, K))
        orig_fwd(ms, Zs, TZs, K)

    def spy_bwd(ms, Gs, K, sm=0):
        records.append(('bwd', ms, tuple(g.shape[2] for g in Gs), bool(sm), False, K))      # (has_p slot: the planes' layout)
        orig_bwd(ms, Gs, K, sm)
    qmesh.spmm2 = ops.spmm2 = spy_spmm
    ops.clip_planes, ops.clip_clenshaw = spy_fwd, spy_bwd
    try:
        # forward + loss + backward only: no all-reduce, no optimizer step (rank 0 runs this alone)
        nfp.zero_grad()
        nfp.forward_loss(*batch, mask).backward()
        nfp.zero_grad()
        torch.cuda.synchronize()
    finally:
        qmesh.spmm2 = ops.spmm2 = orig_spmm
        ops.clip_planes, ops.clip_clenshaw = orig_fwd, orig_bwd
    dev = batch[0].device
    bufs, timed = {}, {}
    side = torch.cuda.Stream()
    tot = {'us': 0.0, 'bytes': 0.0, 'hop_bytes': 0.0, 'moved': 0.0, 'hops': 0}
    per_kind = {}
    for kind, ms, Cs, has_p, has_q, K in records:
        key = (kind, id(ms), Cs, has_p, has_q, K)
        if key not in timed:
            if kind == 'hop':
                bk = (ms.N, Cs)
                if bk not in bufs:
                    bufs[bk] = [[torch.randn(ms.N, c, device=dev) for c in Cs] for _ in range(4)]
                x, p, q, out = bufs[bk]
                fn = lambda: orig_spmm(ms, x, 2.0, p if has_p else None, -1.0, q if has_q else None, 1.0, out)
            elif kind == 'fwd':
                Zs = [torch.randn(ms.N, c, device=dev) for c in Cs]
                TZs = [torch.empty(K - 1, ms.N, c, device=dev) for c in Cs]
                fn = lambda: orig_fwd(ms, Zs, TZs, K)
            else:
                Gs = [torch.randn(K, ms.N, c, device=dev) for c in Cs]
                fn = lambda: orig_bwd(ms, Gs, K, int(has_p))
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                for _ in range(reps):
                    fn()
            g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            timed[key] = a.elapsed_time(b) * 1e3 / reps
            del g
        us = timed[key]
        nv = ms.n_valid                      # static mode: ms.N is the capacity, the count lives on the device
        C = sum(Cs)
        idx = 4.0 * (nv + 1) + 8.0 * ms.E
        hop = idx + 8.0 * nv * C
        if kind == 'hop':
            by, hops = hop, 1
            moved = hop + 4.0 * nv * C * (int(has_p) + int(has_q))
        elif kind == 'fwd':
            by, hops = idx + 4.0 * nv * C * K, K - 1
            moved = by
        else:
            by, hops = idx + 4.0 * nv * C * (K + 1), K - 1
            moved = by
        tot['us'] += us
        tot['bytes'] += by
        tot['hop_bytes'] += hop * hops
        tot['moved'] += moved
        tot['hops'] += hops
        pk = per_kind.setdefault(kind, {'launches': 0, 'us': 0.0, 'bytes': 0.0})
        pk['launches'] += 1
        pk['us'] += us
        pk['bytes'] += by
    n = len(records)
    # `achieved` / `frac`: the bytes a launch MUST move through HBM (algorithmic, per launch) / its measured duration.  A fused
    # launch is priced on its own compulsory bytes -- index arrays + Z once + K - 1 planes out (backward: K planes in, one
    # out) --, NOT on hops x the per-hop figure: the intermediate planes of a fused recurrence never cross HBM, and crediting
    # them overstated the fraction by 1.4 - 1.65 x (round-3 advisor finding).  The per-hop pricing stays as
    # `equivalent_unfused` (what the same hops would have moved as K - 1 separate launches: comparable with rounds 1 - 2).
    unfused = tot['hop_bytes'] / (tot['us'] * 1e-6) / 1e9
    achieved = tot['bytes'] / (tot['us'] * 1e-6) / 1e9
    # HBM-side bytes per launch come from separate rocprofv3 --pmc passes over this same command (they cannot be collected
    # in-process); the newest committed record is quoted and named
    traffic = source = None
    for name in traffic_files:
        pmc = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(pmc):
            traffic, source = json.load(open(pmc))['traffic_bytes_per_launch'], 'profiles/' + name
            break
    fused = per_kind.get('fwd', {'launches': 0})['launches'] + per_kind.get('bwd', {'launches': 0})['launches']
    kinds = {('k_cheb_clip<fwd>' if k == 'fwd' else 'k_cheb_clip<bwd>' if k == 'bwd' else 'k_spmm'):
             {'launches_per_step': v['launches'], 'avg_launch_us': round(v['us'] / v['launches'], 2),
              'avg_bytes_per_launch': round(v['bytes'] / v['launches']),
              'achieved_gbs': round(v['bytes'] / (v['us'] * 1e-6) / 1e9, 1)} for k, v in per_kind.items()}
    rec = {'bound': 'hbm', 'kernel': 'k_cheb_clip (clip-resident multi-hop message aggregate)' if fused else 'k_spmm',
           'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': source,
           'launches_per_step': n, 'hops_per_step': tot['hops'], 'avg_launch_us': round(tot['us'] / n, 2),
           'us_per_step': round(tot['us'], 1), 'avg_bytes_per_launch': round(tot['bytes'] / n),
           'avg_us_per_hop': round(tot['us'] / max(tot['hops'], 1), 2), 'kernels': kinds,
           'equivalent_unfused': {'achieved': round(unfused, 1), 'frac': round(unfused / HBM_PEAK_GBS, 4),
                                  'avg_bytes_per_launch': round(tot['hop_bytes'] / n),
                                  'note': "SURVEY 8(d)'s per-hop figure x the hops a launch processes: what K - 1 separate "
                                          'launches would have moved; a fused launch never re-reads its intermediate planes, so '
                                          'this is NOT HBM traffic (rounds 1-3 reported it as `frac`)'},
           'bytes_formula': "per launch: k_spmm (one hop) 4(N+1) + 8E' + 8NC (SURVEY 8(d)); fused K-1 hops: forward "
                            "4(N+1) + 8E' + 4NC K, backward 4(N+1) + 8E' + 4NC (K+1)"}
    # context, not a roof: what a plain streaming launch (torch.add: two operands read, one written) that moves the average launch's
    # bytes takes under the SAME protocol (`reps` launches back to back in one graph: operands warm in the 256 MB Infinity Cache,
    # like the recurrence's, whose input the previous launch of the step wrote) -- tools/exp_ceiling.py has the cold figures
    nel = max(int(tot['bytes'] / n / 12), 1)
    ca, cb, cc = (torch.randn(nel, device=dev) for _ in range(3))
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.add(ca, cb, out=cc)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        for _ in range(reps):
            torch.add(ca, cb, out=cc)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    b.synchronize()
    copy_us = a.elapsed_time(b) * 1e3 / reps
    del g
    rec['plain_streaming_launch_same_bytes'] = {
        'us': round(copy_us, 2), 'gbs': round(12.0 * nel / (copy_us * 1e-6) / 1e9, 1),
        'kernel_time_over_this': round(tot['us'] / n / copy_us, 2),
        'note': 'torch.add on operands of the average launch\'s bytes, same replay protocol; the practical ceiling of a launch this '
                'short, not the 8 TB/s roof that `frac` is priced against'}
    if fused:
        rec['limiter'] = ('not HBM: one CU\'s vector issue + LDS per workgroup (SQ counters: profiles/r05_pmc_clip_sq.json); '
                          '`bound` names the roof the contract prices against')
    if traffic:
        rec['traffic_gbs'] = round(traffic / (tot['us'] / n * 1e-6) / 1e9, 1)
    return rec


def rollout_sizes(nfp, batch, mask):
    """N_t and E'_t (valid nodes, directed non-self edges) of the input mesh and of the mesh every decoder step runs
    on, for one forward pass of `batch` on the model as it is -- so that the bytes of a step can be recomputed from the per-step
    sizes and the formulas of DESIGN.md section 5 instead of from launch averages."""
    import torch
    with torch.no_grad():
        _, meshes = nfp.model(batch[0], batch[1], batch[2], teacher_forcing_ratio=0, mask=mask)
    out = []
    for t, ms in enumerate(meshes):
        out.append({'step': t, 'N': int(ms.n_valid), 'E': int(ms.E), 'mesh': 'input' if t == 0 else f're-mesh after step {t - 1}'})
    return {'clips': int(meshes[0].B), 'pixels_per_clip': int(meshes[0].P), 'meshes': out,
            'note': "mesh of decoder step t (step 0 runs on the input mesh, which also serves the T_in encoder steps); N = nodes of the "
                    "block-diagonal batch graph, E = directed edges without self pairs (E' of SURVEY 8(d))"}


def attn_roofline(nfp, batch, mask):
    """Roofline object of the attention launches (TransformerConv configurations; tools/bench_configs.py): every qt_attn_fwd /
    qt_attn_bwd call of one eager forward + backward is bracketed by HIP events on the launch stream (a qt_attn_bwd call is the
    target pass + the source pass, two kernels).  Algorithmic bytes = the operands once (G heads, N nodes, C channels, E' edges;
    DESIGN.md section 5): forward 20 G N C (q, k, v, skip read; out written) + 8 G N (softmax stats) + index arrays; backward
    44 G N C (target pass: g, q, k, v, out read, dq and dskip written = 28; source pass: g, q read, dk, dv written = 16)
    + 16 G (E' + N) (the per-message coefficients written and read once) + index arrays twice."""
    import torch
    from qtmpnn import _lib
    events, orig = [], _lib.call

    def spy(name, *args):
        if name in ('qt_attn_fwd', 'qt_attn_bwd'):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            orig(name, *args)
            b.record()
            if name == 'qt_attn_fwd':      # (rowptr, col, xy, eattr, self, proj, ld, We, C, c_real, N, n_dev, ..., G at 17)
                C, N, G = args[8], args[10], args[17]
            else:                          # (..., C at 8, N at 10, ..., E at 25, G at 26)
                C, N, G = args[8], args[10], args[26]
            events.append((name, a, b, G, N, C))
        else:
            orig(name, *args)
    from qtmpnn import ops
    _lib.call = spy
    try:
        nfp.zero_grad()
        loss = nfp.forward_loss(*batch, mask)
        mesh0 = nfp.model.graph.mapping
        loss.backward()
        nfp.zero_grad()
        torch.cuda.synchronize()
    finally:
        _lib.call = orig
    nv, E = mesh0.n_valid, mesh0.E
    idx = 4.0 * (nv + 1) + 12.0 * E + 12.0 * nv            # rowptr, col, [angle, dist] per edge, centroids, self-pair flags
    tot = {}
    for name, a, b, G, N, C in events:
        us = a.elapsed_time(b) * 1e3
        if name == 'qt_attn_fwd':
            by = 20.0 * G * nv * C + 8.0 * G * nv + idx
        else:
            by = 44.0 * G * nv * C + 16.0 * G * (E + nv) + 8.0 * G * nv + 2 * idx + 4.0 * E
        d = tot.setdefault(name, {'launches': 0, 'us': 0.0, 'bytes': 0.0})
        d['launches'] += 1
        d['us'] += us
        d['bytes'] += by
    us = sum(d['us'] for d in tot.values())
    by = sum(d['bytes'] for d in tot.values())
    n = sum(d['launches'] for d in tot.values())
    ach = by / (us * 1e-6) / 1e9
    return {'bound': 'hbm', 'kernel': 'k_attn_fwd / k_attn_bwd_target / k_attn_bwd_source (edge-softmax attention, 8 heads per launch)',
            'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': None,
            'launches_per_step': n, 'us_per_step': round(us, 1), 'avg_launch_us': round(us / n, 2), 'avg_bytes_per_launch': round(by / n),
            'kernels': {k: {'launches_per_step': d['launches'], 'avg_launch_us': round(d['us'] / d['launches'], 2),
                            'avg_bytes_per_launch': round(d['bytes'] / d['launches']),
                            'achieved_gbs': round(d['bytes'] / (d['us'] * 1e-6) / 1e9, 1)} for k, d in tot.items()},
            'timing': 'HIP events on the launch stream around every call of one eager forward + backward (qt_attn_bwd = target + source pass)',
            'bytes_formula': "forward 20 G N C + 8 G N + idx; backward 44 G N C + 16 G (E' + N) + 8 G N + 2 idx + 4 E'; "
                             "idx = 4 (N + 1) + 12 E' + 12 N"}


def step_record(ms_per_step):
    """The whole-step record (profiles/r05_step_summary.json: launches, launch-floor tail and HBM-side bytes of one replayed step
    from a rocprofv3 kernel trace + the PMC traffic passes of the same build) with the rate THIS run's step time implies."""
    for name in ('r05_step_summary.json',):
        f = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(f):
            rec = json.load(open(f))
            rec = {k: rec[k] for k in ('launches', 'launches_lt_10us', 'launches_lt_10us_ms', 'hbm_bytes', 'hbm_gb',
                                       'floor_ms_at_copy_rate', 'copy_rate_tbs') if k in rec}
            rec['implied_tbs'] = round(rec['hbm_bytes'] / (ms_per_step * 1e-3) / 1e12, 2)
            rec['frac_of_hbm_peak'] = round(rec['implied_tbs'] * 1e3 / HBM_PEAK_GBS, 3)
            rec['source'] = 'profiles/' + name + ' (bytes, launch counts: rocprofv3 passes of the same build); rate: this run'
            return rec
    return None


MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (= the fp32 vector rate)


def gemm_mfma(nfp, batch, mask, reps=20):
    """MFMA utilisation of the gate GEMM (north_star): the encoder's layer-0 launch of this workload -- qt_dense_lstm on
    Z = [X (4) | H (16)], K' = 5 Chebyshev planes + bias rows -> (N x 104)(104 x 64) + the fused LSTM cell -- re-issued `reps`
    times from a hipGraph on the input mesh of the batch, timed with HIP events on the replay stream.  achieved = 2 N K 4h
    flops / launch time against the dense fp32-MFMA peak.  The counter view of the same launch (SQ_VALU_MFMA_BUSY_CYCLES per
    SIMD-cycle, wave stall breakdown) is a separate rocprofv3 --pmc pass: profiles/r02_pmc_gemm.json."""
    import torch
    from qtmpnn import _lib
    from qtmpnn._lib import ptr
    x = batch[0]
    dev = x.device
    nfp.model.eval()
    with torch.no_grad():
        mesh = nfp.model._mesh_from_image(x[..., 0].amax(dim=1), mask, None)
    nfp.model.train()
    N, h, K, Ca, Cab = mesh.N, HIDDEN, 5, 4, HIDDEN
    nv = mesh.n_valid
    Kt = K * (Ca + Cab) + 4
    from qtmpnn import ops
    sm = int(ops._clip_resident(mesh, [Ca, Cab], K))          # the plane layout of the real launch (slice-major from k_cheb_clip)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    X, H, TX, TH = rnd(N, Ca), rnd(N, Cab), rnd(K - 1, N, Ca), rnd(K - 1, N, Cab)
    S = mesh.cheb_ones(3)
    W = 0.1 * rnd(Kt, 4 * h)
    WT = W.t().contiguous()
    Cp, wc, b, ln = rnd(N, h), 0.1 * rnd(3, h), 0.1 * rnd(4, h), rnd(4, h)
    Hn, Cn, gates = (torch.empty(N, w, device=dev) for w in (h, h, 4 * h))
    fn = lambda: _lib.call('qt_dense_lstm', ptr(X), Ca, ptr(TX), ptr(H), Cab, ptr(TH), K, Ca, Cab, ptr(W), ptr(WT), ptr(S), 4,
                           ptr(W[K * (Ca + Cab):]), h, N, ptr(mesh.n_dev), ptr(Cp), h, ptr(wc), ptr(b), ptr(ln), None, ptr(Hn),
                           ptr(Cn), ptr(gates), sm)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side, capture_error_mode='thread_local'):
        for _ in range(reps):
            fn()
    gr.replay()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    gr.replay()
    e.record()
    e.synchronize()
    us = a.elapsed_time(e) * 1e3 / reps
    tflops = 2.0 * nv * Kt * 4 * h / us / 1e6
    rec = {'kernel': 'k_gate_cell_p (qt_dense_lstm: gate GEMM + LSTM cell)', 'shape': f'({nv} x {Kt})({Kt} x {4 * h})',
           'launch_us': round(us, 2), 'achieved_tflops': round(tflops, 1), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(tflops / MFMA_F32_PEAK_TFLOPS, 3),
           'hbm_gbs': round(4.0 * nv * (Kt + 7 * h) / us / 1e3, 1)}
    for name in ('r05_pmc_gemm.json', 'r04_pmc_gemm.json', 'r03_pmc_gemm.json', 'r02_pmc_gemm.json'):
        pmc = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(pmc):
            rec['counters'] = {k: v for k, v in json.load(open(pmc)).items() if k in ('mfma_busy_frac', 'wave_stall_frac', 'source')}
            rec['counters']['file'] = 'profiles/' + name
            break
    return rec


def cpu_baseline(n_clips, n_warm=3):
    """The CPU oracle (kind "port": a restatement of the reference algorithm, see oracle/qt_oracle.py) on a bounded sample,
    as SURVEY.md 8(d) prescribes: the cfg1 clips (64x64, ONE digit, in=10/out=10, noise 0.05), one clip per optimizer step
    like the reference's batch_size=1 loop, `n_warm` warm-up clips, then the MEDIAN clip time of `n_clips` clips.  Threads =
    the cores this process may use, at most 16: a one-GPU job owns a 16-core share of the box, and with one thread per
    visible core (128+) the small per-node tensor ops only thrash (measured: > 40 s per clip instead of ~1.5 s);
    `cores` reports the thread count actually used."""
    import numpy as np
    import torch
    from oracle import qt_oracle as O
    from qtmpnn import synthetic
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(cores, 16)))
    torch.manual_seed(1)
    model = O.Seq2Seq(HIDDEN, DROPOUT, THRESH, input_timesteps=T_IN, input_features=4, output_timesteps=T_OUT,
                      n_layers=N_LAYERS, n_conv_layers=2)
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    mask = np.zeros(CANVAS, dtype=bool)
    x, y = synthetic.make_batch(1, 0, n_warm + n_clips, T_IN, T_OUT, n_digits=1, pixel_noise=NOISE, canvas=CANVAS)
    concat = torch.zeros(T_OUT, *CANVAS, 1)
    times = []
    for i in range(n_warm + n_clips):
        t0 = time.perf_counter()
        O.train_step(model, opt, torch.from_numpy(x[i]), torch.from_numpy(y[i]), concat, mask)
        times.append(time.perf_counter() - t0)
        log(f'cpu baseline clip {i + 1}/{n_warm + n_clips}: {times[-1]:.2f} s')
        if sum(times) > 120 and i >= n_warm + 2:          # bounded sample: never more than ~2 minutes of CPU work
            break
    n_clips = len(times) - n_warm
    med = float(np.median(times[n_warm:]))
    return {'value': round((T_IN + T_OUT) / med, 3), 'unit': 'frames/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'sample': f'cfg1 clips (64x64, 1 digit, in=10/out=10, noise {NOISE}), one clip per optimizer step: '
                                      f'median of {n_clips} clips after {n_warm} warm-up clips ({sum(times):.1f} s in all)'}


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started by hand: become the torchrun parent BEFORE anything touches the GPU
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29511'),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # stdout carries ONE JSON line and nothing else: RCCL prints its version banner to file descriptor 1 when the first
    # communicator is created (seen with --force-multi), so fd 1 is pointed at stderr for the run and the line is written to
    # the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from qtmpnn import ops, synthetic
    from qtmpnn.dist import HostBarrier, broadcast_parameters, init_from_env

    # QT_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box); the real runs use RCCL
    rank, world, local = init_from_env(os.environ.get('QT_DIST_BACKEND', 'nccl'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    side = HostBarrier()                 # (collective: every rank, right after init)
    device = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if args.force_multi:
        assert world == 1 and not args.eager, '--force-multi is the one-GPU rehearsal of the multi-rank step (hipGraph replay)'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29513')
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
    assert not ops.DGRAD_SPLIT_BF16, 'the headline is the exact-fp32 step: unset QT_DGRAD_SPLIT_BF16'
    nfp = make_predictor(device, capturable=not args.eager)
    if world > 1:
        broadcast_parameters(nfp.model)
        torch.manual_seed(1000 + rank)      # same weights everywhere, but every rank draws its own dropout masks
    nfp.model.train()

    # synthetic batches resident in HBM before the timed region; rank r owns clips [r*B, (r+1)*B) of each batch
    mask = np.zeros(CANVAS, dtype=bool)
    n_pool = 4
    pool = []
    for i in range(n_pool):
        x, y = synthetic.make_batch(2, (i * world + rank) * args.batch, args.batch, T_IN, T_OUT, n_digits=N_DIGITS,
                                    pixel_noise=args.noise, canvas=CANVAS)
        pool.append((torch.from_numpy(x).to(device), torch.from_numpy(y).to(device),
                     torch.zeros(args.batch, T_OUT, *CANVAS, 1, device=device)))
    host_pool = [tuple(t.cpu().pin_memory() for t in b) for b in pool] if args.host_inputs else None

    log(f'rank {rank}: {n_pool} batches of {args.batch} clips resident on {device}')

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def host_barrier():
        """All ranks meet on the HOST (gloo side group created at init, qtmpnn.dist.HostBarrier): a rank that waits here for
        rank 0's seconds of single-rank work sits in no RCCL collective (nothing for the watchdog to time, no GPU spin)."""
        side.wait()

    def set_lr(opt, v):
        g = opt.param_groups[0]
        if torch.is_tensor(g['lr']):
            g['lr'].fill_(v)
        else:
            g['lr'] = v

    def timed(step, n):
        """n steps bracketed by barrier + synchronize on both sides; max over ranks of the wall time.  Also returns every
        rank's own time to finish its n steps (before the closing barrier): load imbalance from data-dependent meshes."""
        barrier()
        t0 = time.perf_counter()
        for i in range(n):
            if host_pool is not None:
                loss = step(*(t.to(device, non_blocking=True) for t in host_pool[i % n_pool]))
            else:
                loss = step(*pool[i % n_pool])
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        per_rank = [own]
        if world > 1:
            tmax = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
            owns = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(owns, torch.tensor([own], device=device, dtype=torch.float64))
            per_rank = [float(o.item()) for o in owns]
        return dt, loss, per_rank

    def frozen_ms(p, step, n):
        """ms per step of `step` with the learning rate at 0 (three untimed steps first); Adam's state is zeroed afterwards.
        The caller sets the learning rate to 0 BEFORE make_graphed_step, whose two eager warm-up steps are real optimizer steps:
        the frozen model is then the model as initialised, the same for every build and every gradient arithmetic (two real
        steps with another rounding of the gradients end in another model, other meshes and another step time)."""
        lr = float(os.environ.get('QT_BENCH_LR', LR))
        set_lr(p.optimizer, 0.0)
        for i in range(3):
            step(*pool[i % n_pool])
        dtf, _, _ = timed(step, n)
        set_lr(p.optimizer, lr)
        for st in p.optimizer.state.values():         # the frozen steps leave no trace in Adam's moments / step counts
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        return dtf / n * 1e3

    # Phase 0 (untimed for `value`): the same step with the learning rate at 0.  The decoder's meshes follow the model's own
    # output, so the trained number below moves with the training trajectory; the frozen one is comparable between builds.
    if args.eager:
        step = lambda x, y, c: nfp.train_step(x, y, c, mask)
    else:
        # the whole step (forward, loss, backward, clip, Adam) as ONE hipGraph; its 2 eager warm-up steps run first (with the
        # learning rate at 0 when the frozen phase follows: see frozen_ms)
        if args.frozen_steps > 0:
            set_lr(nfp.optimizer, 0.0)
        step = nfp.make_graphed_step(*pool[0], mask=mask, warmup=2, force_multi=args.force_multi)
        log('training step captured into a hipGraph')
    frozen = None
    if args.frozen_steps > 0 and args.eager:
        set_lr(nfp.optimizer, 0.0)
    if args.frozen_steps > 0:
        frozen = frozen_ms(nfp, step, args.frozen_steps)
        log(f'frozen model (lr = 0): {frozen:.3f} ms per step over {args.frozen_steps} steps')

    for i in range(args.warmup):
        l = step(*pool[i % n_pool])
        if rank == 0:
            log(f'warm-up step {i}: loss {float(l):.5f}')
    dt, loss, per_rank = timed(step, args.steps)
    assert torch.isfinite(loss).item(), 'non-finite loss in the timed region'
    final_loss = float(loss)
    log(f'timed {args.steps} steps in {dt:.3f} s')
    # further windows of the same length (training goes on, so they are not the same steps: the spread shows how much of a
    # difference between two runs is the box / the trajectory and how much is the build)
    windows = [dt / args.steps * 1e3]
    for _ in range(max(args.repeats, 0)):
        dtr, lr_, _ = timed(step, args.steps)
        assert torch.isfinite(lr_).item(), 'non-finite loss in a repeat window'
        windows.append(dtr / args.steps * 1e3)

    # ---- single-rank work AFTER every timed window.  The other ranks wait on the host (store barrier), not in a collective.
    probes = {}
    if rank == 0:
        if not args.no_roofline or (not args.no_split_variant and not args.eager):
            del step
            nfp._graph = None
            fresh = make_predictor(device, capturable=False)      # the model as initialised: the same launch mix in every build
            fresh.model.train()
        if not args.no_roofline:
            try:
                probes['roofline'] = spmm_roofline(fresh, pool[0], mask)
                m = gemm_mfma(fresh, pool[0], mask)
                if m is not None:
                    probes['mfma'] = m
                probes['rollout_sizes'] = rollout_sizes(fresh, pool[0], mask)
            except Exception as e:                                            # pragma: no cover
                probes.setdefault('roofline', {'error': repr(e)[:200]})
            log('roofline probes done')
        if not args.no_split_variant and not args.eager and args.frozen_steps > 0 and world == 1:
            # the same captured step with the opt-in split-bf16 data gradient (2-term bf16 split on the bf16 MFMA, fp32
            # accumulate; gradients only): frozen model, same box, same invocation -- beside the exact-fp32 headline
            try:
                prev = ops.set_dgrad_split_bf16(True)
                sp = make_predictor(device, capturable=True)
                sp.model.train()
                set_lr(sp.optimizer, 0.0)
                sstep = sp.make_graphed_step(*pool[0], mask=mask, warmup=2)
                sf = frozen_ms(sp, sstep, args.frozen_steps)
                probes['split_bf16_dgrad'] = {
                    'frozen_ms_per_step': round(sf, 3), 'exact_fp32_frozen_ms_per_step': round(frozen, 3),
                    'frozen_value': round(args.batch * (T_IN + T_OUT) / sf * 1e3, 1),
                    'what': 'opt-in (QT_DGRAD_SPLIT_BF16=1): backward gate-GEMM data gradient as hi*hi + hi*lo + lo*hi on '
                            'v_mfma_f32_32x32x16_bf16, fp32 accumulate; NOT part of `value`'}
                del sstep, sp
            except Exception as e:                                            # pragma: no cover
                probes['split_bf16_dgrad'] = {'error': repr(e)[:200]}
            finally:
                ops.set_dgrad_split_bf16(prev)
            log('split-bf16 variant timed')
    host_barrier()

    if rank == 0:
        global_batch = args.batch * world
        frames = global_batch * (T_IN + T_OUT) * args.steps
        srt = sorted(windows)
        med_ms = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        line = {
            'metric': 'frames/sec (train fwd+bwd), 64x64 MovingMNIST in=10/out=10', 'value': round(frames / dt, 1),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: Moving-MNIST-like 64x64, 2 digits, in=10/out=10, '
                                   f'{args.batch} clips per GPU, pixel noise {args.noise}, thresh {THRESH}, hidden {HIDDEN}, '
                                   f'{N_LAYERS} layers, ChebConv K=3, dropout {DROPOUT}, Adam',
                       'global_batch': global_batch, 'frames_per_clip': T_IN + T_OUT, 'parallelism': f'dp{world}',
                       'launch': 'eager' if args.eager else 'graph 1 (forward + loss + backward), RCCL all-reduce of the flat '
                                 'gradient in a group of one rank, graph 2 (clip + Adam)' if args.force_multi else 'hipGraph replay',
                       'inputs': 'pinned host memory, copied every step' if args.host_inputs else 'resident in HBM',
                       'arithmetic': 'fp32 throughout: every GEMM (forward, data gradient, weight gradient) on '
                                     'v_mfma_f32_32x32x2_f32; no reduced-precision operand anywhere in the timed step',
                       'final_loss': round(final_loss, 6)},
            'frozen_ms_per_step': None if frozen is None else round(frozen, 3),
            'window_ms_per_step': [round(w, 3) for w in windows],
            'median_ms_per_step': round(med_ms, 3), 'median_value': round(global_batch * (T_IN + T_OUT) / med_ms * 1e3, 1),
            'spread_ms_per_step': round(srt[-1] - srt[0], 3),
            'rank_ms_per_step': {'min': round(min(per_rank) / args.steps * 1e3, 3), 'max': round(max(per_rank) / args.steps * 1e3, 3)},
        }
        line.update(probes)
        st = step_record(dt / args.steps * 1e3)
        if st is not None:
            line['step'] = st
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(args.cpu_clips)
            except Exception as e:                                            # pragma: no cover
                line['cpu_baseline'] = {'error': repr(e)[:200]}
        from qtmpnn.mesh import tile_error_word
        assert tile_error_word() == 0, 'a tile-resident launch reported an error (persistent error word)'
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + '\n').encode())
    if world > 1:
        host_barrier()
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
