"""Benchmark of the walk-training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c5]

One "step" = one full iteration of the reference's training loop (train.py:48-110) over one batch of synthetic z:
sample z -> style MLP -> StyleGAN2 synthesis (original) -> ResNet-50 regressor -> linear W+ walk -> synthesis (edited)
-> discriminator + VGG-19 content + regressor BCE losses -> backward into the walk -> [all-reduce] -> Adam.

--config c3 (default, the headline): BASELINE.json configs[2] — StyleGAN2 FFHQ-shaped 1024^2 generator, ResNet-50 regressor,
  1 attribute (Smiling), batch 8 per GPU, full loss (the reference's default flags), train.py flow, fp32 on the matrix cores
  (the reference's arithmetic type), eager launches on three loss-branch streams.
--config c5: BASELINE.json configs[4] per-GPU shape — the same networks on the transient-scene attribute table (SceneGraph,
  dataset/attributes_scene.txt, 5 attributes, train_multi_attr.py clamp flow), batch 8 per GPU, every contraction on the 16-bit
  matrix cores (3-term bf16 split, fp32 accumulation: fp32-class results), forward+backward replayed from ONE hipGraph.

N > 1: one rank process per GPU (RCCL over xGMI).  `python bench.py --gpus N` starts the ranks itself; under
torch.distributed.run (WORLD_SIZE set) it is one of them.  Weak scaling: the per-GPU batch is fixed; ranks draw identical z /
alpha and take every N-th sample; the only collective is one all-reduce of the walk gradient per step.  Prints ONE JSON line
on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6300 achievable)
NOISE_STRENGTH = 0.05               # per-layer NoiseInjection weights of the benchmarked generator ("pretrained-like": non-zero)
SCENE_ATTRS = ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk']


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(resolution, n_attr, budget_s, full_loss=True, clamp=False):
    """The CPU oracle (plain torch on the host cores) on a bounded sample of the workload, two ways:
    (1) whole training steps at the BENCHMARK resolution with batch 1 (per-image work is identical; D's stddev group is min(B,4)) — `value`;
    (2) >= 3 whole training steps at the REFERENCE's own shape, 256^2 batch 4 (README / constants.py:1 of the reference) — `reference_shape`."""
    from latent2im_amd import synth
    from oracle import step as ostep
    # torch's CPU convs stop scaling (and collapse when oversubscribed: 256 threads on the 256-core box = 334 s per step,
    # 32 threads = 18 s), so the port is timed on at most 32 cores and `cores` reports what was actually used
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)

    def timed(res, batch, min_steps, max_steps, budget):
        nets = dict(G=ostep.to_torch(synth.generator_state(res, seed=100)), D=ostep.to_torch(synth.discriminator_state(res, seed=200)),
                    R=ostep.to_torch(synth.resnet50_state(seed=300)), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400)))
        n_latent = 2 * int(np.log2(res)) - 2
        walk = torch.from_numpy(synth.walk_init(n_attr, n_latent, seed=7))
        done, t_total = 0, 0.0
        while True:
            z = torch.from_numpy(synth.z_sample(batch, seed=done)).float()
            t0 = time.time()
            ostep.train_step(nets, walk, z, torch.full((batch, n_attr), 0.3), list(range(n_attr)),
                             no_content_loss=not full_loss, no_gan_loss=not full_loss, clamp_variant=clamp)
            t_total += time.time() - t0
            done += 1
            if done >= max_steps or (done >= min_steps and t_total >= budget):
                break
        return done, t_total

    done, t_total = timed(resolution, 1, 1, 4, budget_s)
    out = dict(value=done / t_total, unit='images/s', cores=threads, kind='port', cpu=cpu_model(), host_threads=os.cpu_count(),
               sample='%d full training step(s) of batch 1 at %d^2 (same losses), %.1f s of CPU work, torch %s CPU ops'
                      % (done, resolution, t_total, torch.__version__))
    d2, t2 = timed(256, 4, 3, 6, budget_s)
    out['reference_shape'] = dict(value=4 * d2 / t2, unit='images/s', steps=d2, s_per_step=round(t2 / d2, 3),
                                  sample='%d full training steps of batch 4 at 256^2 (the reference\'s own run shape), %.1f s of CPU work' % (d2, t2))
    return out


def lib_hash():
    from latent2im_amd import _lib
    with open(_lib.LIB_PATH, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def call_bytes(q):
    """Algorithmic HBM bytes of a conv call: input (+ its gradient mask) read once, output written once, every fused epilogue
    operand (residual, its mask, output mask, accumulate, res_sub) read once; weights / per-sample vectors are noise."""
    B, cin, cout, kh, kw, stride, H, W, OH, OW, step, in_mask = q[3][:12]
    flags = q[3][13] if len(q[3]) > 13 else ''
    extra = sum(1 for c in 'rmoas' if c in flags.rstrip('0123456789'))
    bpe = 2.0 if q[5].endswith('_h8') else 4.0              # bf16 h8 maps on the 16-bit path
    npx_out = (4 * H * W) if (q[5] == 'transposed_h8') else OH * OW
    return bpe * (B * cin * H * W * (2 if in_mask else 1) + B * cout * npx_out * (1 + extra))


def family_table(prof, steps):
    """Per kernel family: launches, HIP-event time, algorithmic and executed TFLOP/s, fraction of the peak of the MFMA instruction
    the family runs on.  executed = what the matrix cores actually multiply (Winograd F(2x2,3x3): 16/36 of the dense form; the
    bf16 split: 3 bf16 products per fp32 product)."""
    from latent2im_amd import conv
    fam = {}
    for q in prof:
        f = fam.setdefault(q[5], dict(n=0, ms=0.0, flop=0.0, bytes=0.0))
        f['n'] += 1
        f['ms'] += q[0].elapsed_time(q[1])
        f['flop'] += q[2]
        f['bytes'] += call_bytes(q)
    rows = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
        kern, exe_ratio, peak = conv.FAMILY_INFO[name]
        alg = f['flop'] / (f['ms'] * 1e-3) / 1e12
        rows.append(dict(family=name, kernel=kern, launches_per_step=round(f['n'] / steps, 1), ms_per_step=round(f['ms'] / steps, 3),
                         avg_launch_ms=round(f['ms'] / f['n'], 4), algorithmic_tflops=round(alg, 2), executed_tflops=round(alg * exe_ratio, 2),
                         peak_tflops=peak, frac=round(alg * exe_ratio / peak, 4),
                         algorithmic_bytes_per_launch=round(f['bytes'] / f['n']), algorithmic_flop_per_launch=round(f['flop'] / f['n']),
                         hbm_GBs_algorithmic=round(f['bytes'] / (f['ms'] * 1e-3) / 1e9, 1)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c3', choices=['c3', 'c5'], help='BASELINE.json configs[2] (headline) or configs[4] per-GPU shape')
    ap.add_argument('--resolution', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch')
    ap.add_argument('--attrs', type=str, default=None)
    ap.add_argument('--reg_only', action='store_true', help='--no_content_loss --no_gan_loss')
    ap.add_argument('--cpu_baseline_s', type=float, default=12.0, help='CPU-oracle time budget (0 = skip)')
    ap.add_argument('--no_kernel_events', action='store_true', help='do not bracket conv launches with events')
    ap.add_argument('--precision', default=None, choices=['f32', 'bf16x3', 'bf16'],
                    help="f32: exact fp32 MFMA, fp32 storage (c3 default); bf16: the 16-bit path — bf16 h8 storage, one bf16 MFMA per MAC, fp32 accumulation (c5 "
                         "default); bf16x3: fp32 storage, 3-term bf16 split on the matrix cores (fp32-class results)")
    ap.add_argument('--hip_graph', type=int, default=None, help='1: replay forward+backward from one hipGraph (c5 default), 0: eager launches')
    ap.add_argument('--no_alt_precision', action='store_true', help='skip the extra timing of the other matrix path')
    ap.add_argument('--direct_3x3', action='store_true', help='run 3x3 stride-1 layers on the direct implicit-GEMM kernel instead of Winograd F(2x2,3x3)')
    ap.add_argument('--serial_streams', action='store_true', help='run the three loss branches on one stream (profiling aid: with '
                    'concurrent streams the per-kernel durations rocprof reports include time shared with other kernels)')
    ap.add_argument('--dump_launches', type=str, default=None, help='write the per-launch table of the event pass (shape, family, ms, TFLOP/s) to this JSON file')
    ap.add_argument('--noise_strength', type=float, default=NOISE_STRENGTH, help='generator NoiseInjection weights (0: no noise drawn)')
    ap.add_argument('--sweep', type=str, default='1,2,4,8,16', help="batch sweep (BASELINE metric: 'batch sweep'): per-GPU batches timed eagerly after the "
                    "headline, reported as `batch_sweep`; 'none' = off")
    ap.add_argument('--sweep_steps', type=int, default=3)
    ap.add_argument('--rank_timeout_s', type=float, default=3600.0, help='--gpus N without a launcher: give up on the rank processes after this long')
    a = ap.parse_args()

    from latent2im_amd import constants, conv, dist, selfcheck, synth
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL over xGMI) BEFORE anything
        # in this process touches the GPU, relay rank 0's JSON line, fail if any rank failed.  (Under torch.distributed.run the
        # environment already carries WORLD_SIZE and this branch is skipped.)
        codes, out0 = dist.spawn_local(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout=a.rank_timeout_s)
        lines = [l for l in out0.splitlines() if l.startswith('{')]
        if any(codes) or len(lines) != 1:
            sys.stderr.write('bench.py: rank exit codes %s, rank 0 printed %d JSON line(s)\n%s\n' % (codes, len(lines), out0[-2000:]))
            raise SystemExit(1)
        print(lines[0], flush=True)
        return
    c5 = a.config == 'c5'
    precision = a.precision or ('bf16' if c5 else 'f32')
    use_graph = bool(a.hip_graph) if a.hip_graph is not None else c5
    transform = 'scene' if c5 else 'face'
    attrs = a.attrs.split(',') if a.attrs else (SCENE_ATTRS if c5 else ['Smiling'])
    clamp = c5                                              # train_multi_attr.py:113 flow for the multi-attribute configs
    if a.serial_streams:
        constants.CONCURRENT_LOSS_BRANCHES = False
    constants.SYNTH_NOISE_STRENGTH = a.noise_strength
    conv.PRECISION = precision
    conv.USE_WINOGRAD = not a.direct_3x3
    rk, world, local = dist.init_from_env()
    if a.gpus != world:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X'
    dev = torch.device('cuda', torch.cuda.current_device())
    np.random.seed(1234)
    g = selfcheck.build_graph(a.resolution, attrs, a.batch * world, lr=1e-4, transform=transform)
    if world > 1:
        dist.broadcast_parameters(g.walk.parameters())
    flags = dict(no_content_loss=a.reg_only, no_gan_loss=a.reg_only)
    global_b = a.batch * world
    zs_all = synth.z_sample(global_b * (a.steps + a.warmup), seed=0)
    sl = dist.shard(global_b)

    def draw_alpha():                                       # Face/SceneTransform.get_train_alpha: one draw per step, shared by the batch
        lo = -1.0 if c5 else 0.0
        return np.ones((a.batch, len(attrs))) * np.random.uniform(lo, 1, len(attrs))

    captured = None

    def one_step(i):
        zs = zs_all[i * global_b:(i + 1) * global_b][sl]
        if captured is not None:
            return captured(zs, draw_alpha())
        return selfcheck.run_step(g, zs, draw_alpha(), clamp=clamp, **flags)

    def make_captured():
        from latent2im_amd import capture
        return capture.CapturedStep(g, a.batch, len(attrs), clamp=clamp, **flags)

    if use_graph:
        captured = make_captured()
    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        r = one_step(a.warmup + i)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [round(t / a.steps * 1e3, 3) for t in dist.gather_floats(elapsed)]
    elapsed = dist.max_over_ranks(elapsed, dev)
    loss_value = float(r['loss'])
    # multi-GPU evidence that reports itself: which devices the ranks really ran on, and the latency of the step's only collective
    ranks = dist.ranks_seen()
    wgrad = next(iter(g.walk.parameters()))
    allreduce_us = dist.time_allreduce(wgrad.detach())

    # roofline of the conv kernels: the same steps again, eager, with a HIP event pair around every conv launch on the launch
    # stream.  Kept out of the timed region above because ~280 event pairs per step cost ~7 % wall on their own.
    prof, t_events = None, None
    if not a.no_kernel_events:            # every rank repeats the steps (the walk-gradient all-reduce is inside a step); rank 0 reports
        keep_captured, captured = captured, None
        concurrent, constants.CONCURRENT_LOSS_BRANCHES = constants.CONCURRENT_LOSS_BRANCHES, False   # one stream: durations do not overlap
        conv.PROFILE = []
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.steps):
            one_step(a.warmup + i)
        torch.cuda.synchronize()
        t_events = (time.perf_counter() - t1) / a.steps * 1e3
        prof, conv.PROFILE = conv.PROFILE, None
        constants.CONCURRENT_LOSS_BRANCHES = concurrent
        captured = keep_captured
    dist.barrier()

    # batch sweep (the BASELINE metric is quoted over a batch sweep): the same eager step at other per-GPU batches, every rank
    sweep = None
    if a.sweep and a.sweep not in ('0', 'none', 'off'):
        import gc
        captured = None
        gc.collect()
        torch.cuda.empty_cache()
        sweep = []
        for bs in [int(x) for x in a.sweep.split(',') if x]:
            gb = bs * world
            zs_s = synth.z_sample(gb * (a.sweep_steps + 1), seed=1)
            sl_s = dist.shard(gb)

            def sweep_step(i):
                zs = zs_s[i * gb:(i + 1) * gb][sl_s]
                lo = -1.0 if c5 else 0.0
                return selfcheck.run_step(g, zs, np.ones((bs, len(attrs))) * np.random.uniform(lo, 1, len(attrs)), clamp=clamp, **flags)
            try:
                torch.cuda.reset_peak_memory_stats()
                sweep_step(0)
                torch.cuda.synchronize()
                dist.barrier()
                t3 = time.perf_counter()
                for i in range(a.sweep_steps):
                    sweep_step(1 + i)
                torch.cuda.synchronize()
                dist.barrier()
                t_s = dist.max_over_ranks(time.perf_counter() - t3, dev)
                sweep.append(dict(batch=bs, global_batch=gb, images_s=round(gb * a.sweep_steps / t_s, 3), ms_per_step=round(t_s / a.sweep_steps * 1e3, 2),
                                  peak_mem_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1)))
            except torch.OutOfMemoryError as e:
                sweep.append(dict(batch=bs, global_batch=gb, images_s=None, error='out of memory: %s' % str(e)[:120]))
            gc.collect()
            torch.cuda.empty_cache()

    # the other matrix path, same steps, for the record (all ranks: the all-reduce is inside the step)
    alt = None
    if not a.no_alt_precision:
        other = {'f32': 'bf16x3', 'bf16x3': 'f32', 'bf16': 'bf16x3'}[precision]
        conv.PRECISION = other
        if precision == 'bf16':                             # other network classes (fp32 storage): a second graph, the 16-bit one is released first
            import gc
            captured = None
            g = None
            gc.collect()
            torch.cuda.empty_cache()
            np.random.seed(1234)
            g = selfcheck.build_graph(a.resolution, attrs, a.batch * world, lr=1e-4, transform=transform)
            dist.broadcast_parameters(g.walk.parameters())
        if use_graph:                                       # a graph holds the kernels of the precision it was captured with; its memory
            import gc                                       # pool (every activation of a step) is released before the next one is built
            captured = None
            gc.collect()
            torch.cuda.empty_cache()
            captured = make_captured()
        one_step(0)
        torch.cuda.synchronize()
        dist.barrier()
        t2 = time.perf_counter()
        for i in range(a.steps):
            one_step(a.warmup + i)
        torch.cuda.synchronize()
        dist.barrier()
        t_alt = dist.max_over_ranks(time.perf_counter() - t2, dev)
        conv.PRECISION = precision
        alt = dict(precision=other, value=round(global_b * a.steps / t_alt, 3), unit='images/s', ms_per_step=round(t_alt / a.steps * 1e3, 2),
                   note='same workload and steps with the matrix path switched (see DESIGN.md section 2); not the headline value')

    if rk != 0:
        dist.shutdown()
        return
    ms_per_step = elapsed / a.steps * 1e3
    value = global_b * a.steps / elapsed
    roof = None
    if prof and a.dump_launches:
        agg = {}
        for q in prof:
            e = agg.setdefault((q[5],) + tuple(q[3]), [0, 0.0, q[2], call_bytes(q)])
            e[0] += 1
            e[1] += q[0].elapsed_time(q[1])
        rows = [dict(family=k[0], shape=list(k[1:]), launches_per_step=v[0] / a.steps, ms_per_launch=round(v[1] / v[0], 4), ms_per_step=round(v[1] / a.steps, 3),
                     tflops=round(v[2] / (v[1] / v[0] * 1e-3) / 1e12, 1), GBs=round(v[3] / (v[1] / v[0] * 1e-3) / 1e9, 1)) for k, v in agg.items()]
        rows.sort(key=lambda r: -r['ms_per_step'])
        json.dump(rows, open(a.dump_launches, 'w'), indent=0)
    if prof:
        fams = family_table(prof, a.steps)
        dom = fams[0]                                       # the family with the most GPU time per step
        tot_ms = sum(q[0].elapsed_time(q[1]) for q in prof)
        tot_flop = sum(q[2] for q in prof)
        exe_flop = sum(q[2] * conv.FAMILY_INFO[q[5]][1] for q in prof)
        peak_all = max(f['peak_tflops'] for f in fams)
        # HBM bytes of the dominant kernel from the PMC summary that was taken with THIS build of the library (else null)
        traffic, traffic_note = None, 'no PMC summary under profiles/ for this build of libl2i_hip.so and this workload'
        tag = 'c5' if c5 else 'c3'
        tp = os.path.join(ROOT, 'profiles', 'r03_%s_hbm_traffic.json' % tag)
        if os.path.isfile(tp):
            tj = json.load(open(tp))
            ent = tj.get('per_family', {}).get(dom['family'])
            if tj.get('lib_sha256_16') == lib_hash() and tj.get('workload') == [a.resolution, a.batch, attrs, precision, bool(a.reg_only)] and ent \
                    and abs(ent['launches_per_step'] - dom['launches_per_step']) < 0.5:
                traffic = round(ent['bytes_per_launch'])
                traffic_note = ('HBM bytes per launch of %s from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) over this command with '
                                '--serial_streams (profiles/%s, tools/hbm_traffic.py): KiB -> bytes, FETCH_SIZE doubled (gfx950), both factors checked on '
                                'a kernel of known byte count in the same run; the summary carries the sha256 of the library it was taken with and is '
                                'dropped when it differs' % (dom['kernel'], os.path.basename(tp)))
            else:
                traffic_note = 'profiles/%s was taken with another build / workload (library %s now): dropped' % (os.path.basename(tp), lib_hash())
        hbm_frac = round(dom['hbm_GBs_algorithmic'] / HBM_PEAK_GBS, 4)
        if hbm_frac > dom['frac']:                          # the 16-bit kernels: closer to the HBM roof than to the bf16 matrix roof
            head = dict(bound='hbm', kernel=dom['kernel'], family=dom['family'], achieved=dom['hbm_GBs_algorithmic'], peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=hbm_frac, mfma_bound=dict(peak_TFLOPs=dom['peak_tflops'], achieved_TFLOPs_executed=dom['executed_tflops'], frac=dom['frac']))
        else:
            head = dict(bound='mfma', kernel=dom['kernel'], family=dom['family'], achieved=dom['executed_tflops'], peak=dom['peak_tflops'],
                        unit='TFLOP/s', frac=dom['frac'])
        roof = dict(head,
                    traffic=traffic, traffic_unit='bytes per launch', traffic_note=traffic_note,
                    algorithmic_tflops=dom['algorithmic_tflops'], algorithmic_flop_per_launch=dom['algorithmic_flop_per_launch'],
                    algorithmic_bytes_per_launch=dom['algorithmic_bytes_per_launch'], avg_launch_ms=dom['avg_launch_ms'],
                    launches_per_step=dom['launches_per_step'], kernel_ms_per_step=dom['ms_per_step'],
                    hbm_bound=dict(peak_GBs=HBM_PEAK_GBS, achieved_GBs_algorithmic=dom['hbm_GBs_algorithmic'], frac=hbm_frac),
                    families=fams,
                    all_conv=dict(launches_per_step=len(prof) // a.steps, ms_per_step=round(tot_ms / a.steps, 2),
                                  algorithmic_tflop_per_step=round(tot_flop / a.steps / 1e12, 3),
                                  algorithmic_tflops=round(tot_flop / (tot_ms * 1e-3) / 1e12, 2),
                                  executed_tflops=round(exe_flop / (tot_ms * 1e-3) / 1e12, 2),
                                  executed_frac_of_fastest_instruction_peak=round(exe_flop / (tot_ms * 1e-3) / 1e12 / peak_all, 4)),
                    ms_per_step_with_events=round(t_events, 2), library_sha256_16=lib_hash(),
                    note='dominant kernel family = most GPU time per step; bound = the roof it is closer to (hbm: algorithmic bytes of its launches '
                         '/ their HIP-event time against 8 TB/s).  mfma: achieved / frac = FLOPs the matrix cores EXECUTE in that family '
                         '(Winograd F(2x2,3x3): 16/36 of the dense correlation; 3-term bf16 split: 3 products per fp32 product) / its HIP-event time '
                         '/ the dense peak of the MFMA instruction it runs on; algorithmic_tflops = 2*MAC of the dense correlation / the same time. '
                         'Events: an eager repeat of the timed steps on ONE stream with an event pair per conv launch (the timed region has none). '
                         'families: the same figures for every conv kernel family of the step; all_conv: their sum')
    wl = ('StyleGAN2 FFHQ-shaped %d^2 generator (random-init, noise weights ~ %g: per-layer noise drawn every pass), ResNet-50 regressor, '
          '%d %s attr (%s flow), %s, batch %d per GPU, linear W+ walk, %s launches'
          % (a.resolution, a.noise_strength, len(attrs), 'transient-scene' if c5 else 'CelebA', 'train_multi_attr.py clamp' if clamp else 'train.py',
             'reg-only loss' if a.reg_only else 'full loss (reg+content+GAN)', a.batch, 'hipGraph-replayed' if use_graph else 'eager'))
    out = dict(metric='edited images/sec', value=round(value, 3), unit='images/s', n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=round(ms_per_step, 2), higher_is_better=True, scaling='weak', vs_baseline=None, dtype=precision,
               data='synthetic',
               config=dict(workload=wl, baseline_config='configs[4] per-GPU shape' if c5 else 'configs[2]',
                           resolution=a.resolution, global_batch=global_b, per_gpu_batch=a.batch, attrs=attrs,
                           losses='reg' if a.reg_only else 'reg+content+gan', parallelism='dp%d' % world,
                           loss_branch_streams=3 if constants.CONCURRENT_LOSS_BRANCHES else 1, hip_graph=use_graph,
                           noise_strength=a.noise_strength, loss=loss_value),
               roofline=roof, alt_precision=alt, batch_sweep=sweep,
               ranks_seen=ranks, per_rank_ms_per_step=per_rank_ms, allreduce_us=None if allreduce_us is None else round(allreduce_us, 1),
               allreduce_note=('median of 100 synchronised all-reduces of a walk-gradient-shaped tensor (%d bytes) over the %s process group'
                               % (wgrad.numel() * 4, torch.distributed.get_backend()) if allreduce_us is not None
                               else 'no process group at N = 1 (L2I_FORCE_PG=1 builds a one-rank RCCL group as a rehearsal)'))
    if world == 1 and a.cpu_baseline_s > 0:
        out['cpu_baseline'] = cpu_baseline(a.resolution, len(attrs), a.cpu_baseline_s, full_loss=not a.reg_only, clamp=clamp)
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)
    dist.shutdown()


if __name__ == '__main__':
    main()
