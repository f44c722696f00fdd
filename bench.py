"""Benchmark of the walk-training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

One "step" = one full iteration of the reference's training loop (train.py:48-110) over one batch of synthetic z:
sample z -> style MLP -> StyleGAN2 synthesis (original) -> ResNet-50 regressor -> linear W+ walk -> synthesis (edited)
-> discriminator + VGG-19 content + regressor BCE losses -> backward into the walk -> [all-reduce] -> Adam.
Workload = BASELINE.json configs[2]: StyleGAN2 FFHQ-shaped 1024^2 generator, ResNet-50 regressor, 1 attribute
(Smiling), batch 8 per GPU, full loss (the reference's default flags), fp32 on the matrix cores.  Weak scaling: the
per-GPU batch is fixed; ranks draw identical z / alpha and take their slice; the only collective is one all-reduce of
the walk gradient per step.  Prints ONE JSON line on rank 0.
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
import torch

PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same guide: v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK_GBS = 8000.0


def cpu_baseline(resolution, attrs, budget_s, full_loss=True):
    """The CPU oracle (plain torch on the host cores) on a bounded sample of the same workload: whole training steps at
    the benchmark resolution with batch 1 (per-image work is identical; D's stddev group is min(B,4))."""
    from latent2im_amd import synth
    from oracle import step as ostep
    # torch's CPU convs stop scaling (and collapse when oversubscribed: 256 threads on the 256-core box = 334 s per step,
    # 32 threads = 18 s), so the port is timed on at most 32 cores and `cores` reports what was actually used
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    nets = dict(G=ostep.to_torch(synth.generator_state(resolution, seed=100)), D=ostep.to_torch(synth.discriminator_state(resolution, seed=200)),
                R=ostep.to_torch(synth.resnet50_state(seed=300)), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400)))
    n_latent = 2 * int(np.log2(resolution)) - 2
    walk = torch.from_numpy(synth.walk_init(len(attrs), n_latent, seed=7))
    idx = list(range(len(attrs)))
    done, t_total = 0, 0.0
    while True:
        z = torch.from_numpy(synth.z_sample(1, seed=done)).float()
        t0 = time.time()
        ostep.train_step(nets, walk, z, torch.full((1, len(attrs)), 0.3), [31 + i for i in idx],
                         no_content_loss=not full_loss, no_gan_loss=not full_loss)
        t_total += time.time() - t0
        done += 1
        if t_total >= budget_s or done >= 4:
            break
    return dict(value=done / t_total, unit='images/s', cores=threads, kind='port',
                sample='%d full training step(s) of batch 1 at %d^2 (same losses), %.1f s of CPU work, torch %s CPU ops'
                       % (done, resolution, t_total, torch.__version__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--resolution', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch')
    ap.add_argument('--attrs', type=str, default='Smiling')
    ap.add_argument('--reg_only', action='store_true', help='--no_content_loss --no_gan_loss')
    ap.add_argument('--cpu_baseline_s', type=float, default=12.0, help='CPU-oracle time budget (0 = skip)')
    ap.add_argument('--no_kernel_events', action='store_true', help='do not bracket conv launches with events')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16x3'],
                    help="matrix path: exact fp32 MFMA (default) or the opt-in 3-term bf16 split (fp32-class accuracy)")
    ap.add_argument('--no_alt_precision', action='store_true', help='skip the extra timing of the other matrix path')
    ap.add_argument('--direct_3x3', action='store_true', help='run 3x3 stride-1 layers on the direct implicit-GEMM kernel instead of Winograd F(2x2,3x3)')
    ap.add_argument('--serial_streams', action='store_true', help='run the three loss branches on one stream (profiling aid: with\n                    concurrent streams the per-kernel durations rocprof reports include time shared with other kernels)')
    a = ap.parse_args()

    from latent2im_amd import constants, conv, dist, selfcheck, synth
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL over xGMI) BEFORE anything
        # in this process touches the GPU, relay rank 0's JSON line, fail if any rank failed.  (Under torch.distributed.run the
        # environment already carries WORLD_SIZE and this branch is skipped.)
        codes, out0 = dist.spawn_local(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        lines = [l for l in out0.splitlines() if l.startswith('{')]
        if any(codes) or len(lines) != 1:
            sys.stderr.write('bench.py: rank exit codes %s, rank 0 printed %d JSON line(s)\n%s\n' % (codes, len(lines), out0[-2000:]))
            raise SystemExit(1)
        print(lines[0], flush=True)
        return
    if a.serial_streams:
        constants.CONCURRENT_LOSS_BRANCHES = False
    conv.PRECISION = a.precision
    conv.USE_WINOGRAD = not a.direct_3x3
    rk, world, local = dist.init_from_env()
    if a.gpus != world:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X'
    dev = torch.device('cuda', torch.cuda.current_device())
    attrs = a.attrs.split(',')
    np.random.seed(1234)
    g = selfcheck.build_graph(a.resolution, attrs, a.batch * world, lr=1e-4)
    if world > 1:
        torch.distributed.broadcast(g.walk.w.data, src=0)
    flags = dict(no_content_loss=a.reg_only, no_gan_loss=a.reg_only)
    global_b = a.batch * world
    zs_all = synth.z_sample(global_b * (a.steps + a.warmup), seed=0)
    sl = dist.shard(global_b)

    def one_step(i):
        zs = zs_all[i * global_b:(i + 1) * global_b][sl]
        alpha = np.ones((a.batch, len(attrs))) * np.random.uniform(0, 1, len(attrs))      # FaceTransform.get_train_alpha
        return selfcheck.run_step(g, zs, alpha, **flags)

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
    elapsed = dist.max_over_ranks(elapsed, dev)

    # roofline of the dominant kernel: the same steps again, now with a HIP event pair around every conv launch on the
    # launch stream.  Kept out of the timed region above because ~280 event pairs per step cost ~7 % wall on their own.
    prof, t_events = None, None
    if not a.no_kernel_events:            # every rank repeats the steps (the walk-gradient all-reduce is inside a step); rank 0 reports
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
    dist.barrier()

    # the other matrix path, same steps, for the record (all ranks: the all-reduce is inside the step)
    alt = None
    if not a.no_alt_precision:
        other = 'bf16x3' if a.precision == 'f32' else 'f32'
        conv.PRECISION = other
        one_step(0)
        torch.cuda.synchronize()
        dist.barrier()
        t2 = time.perf_counter()
        for i in range(a.steps):
            one_step(a.warmup + i)
        torch.cuda.synchronize()
        dist.barrier()
        t_alt = dist.max_over_ranks(time.perf_counter() - t2, dev)
        conv.PRECISION = a.precision
        alt = dict(precision=other, value=round(global_b * a.steps / t_alt, 3), unit='images/s', ms_per_step=round(t_alt / a.steps * 1e3, 2),
                   note='same workload and steps with the matrix path switched (see DESIGN.md section 2); not the headline value')

    if rk != 0:
        dist.shutdown()
        return
    ms_per_step = elapsed / a.steps * 1e3
    value = global_b * a.steps / elapsed
    roof = None
    if prof:
        # algorithmic HBM bytes of a conv call: input (+ its gradient mask) read once, output written once, every fused
        # epilogue operand (residual, its mask, output mask, accumulate) read once; weights / per-sample vectors are noise
        def call_bytes(q):
            B, cin, cout, kh, kw, stride, H, W, OH, OW, step, in_mask = q[3][:12]
            flags = q[3][13] if len(q[3]) > 13 else ''
            extra = sum(1 for c in 'rmoas' if c in flags.rstrip('0123456789'))
            return 4.0 * (B * cin * H * W * (2 if in_mask else 1) + B * cout * OH * OW * (1 + extra))
        alg_bytes = sum(call_bytes(q) for q in prof) / len(prof)
        traffic, traffic_note = None, 'no PMC summary for this workload under profiles/'
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_conv_hbm_traffic.json')
        if (os.path.isfile(tp) and a.resolution == 1024 and a.batch == 8 and not a.reg_only and a.precision == 'f32' and not a.direct_3x3
                and attrs == ['Smiling']):
            tj = json.load(open(tp))
            if abs(tj['conv_launches_per_step'] - len(prof) // a.steps) < 0.5:
                traffic = round(tj['conv_bytes_per_launch'])
                traffic_note = ('HBM bytes per conv launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command with --serial_streams '
                                '(profiles/r01_conv_hbm_traffic.json, tools/hbm_traffic.py): KiB -> bytes, FETCH_SIZE doubled (gfx950), both factors '
                                'checked on a kernel of known byte count in the same run; not re-measured live (counters need the profiler)')
        tot_ms = sum(q[0].elapsed_time(q[1]) for q in prof)
        tot_flop = sum(q[2] for q in prof)
        ach = tot_flop / (tot_ms * 1e-3) / 1e12
        wino = [q for q in prof if q[4] == 'l2i_conv2d_wino_f32']
        exe_flop = tot_flop - sum(q[2] for q in wino) * (1.0 - 16.0 / 36.0)      # F(2x2,3x3): 16 multiplies per 2x2 tile instead of 36
        exe = exe_flop / (tot_ms * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if a.precision == 'f32' else PEAK_BF16_MFMA_TFLOPS
        roof = dict(bound='mfma', kernel='conv_wino_kernel / conv_mfma_kernel + gemm1x1_kernel / convt_mfma_kernel (l2i_conv2d_wino_f32, l2i_conv2d_f32, l2i_conv_transpose2d_f32)' if a.precision == 'f32'
                    else 'conv_bf16x3_kernel + fp32 kernels for ineligible layers (algorithmic FLOPs; the split executes 3 MFMA FLOPs per algorithmic FLOP)',
                    achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4), traffic=traffic, traffic_unit='bytes per launch', traffic_note=traffic_note,
                    algorithmic_bytes_per_launch=round(alg_bytes),
                    launches_per_step=len(prof) // a.steps, avg_launch_ms=round(tot_ms / len(prof), 4),
                    kernel_ms_per_step=round(tot_ms / a.steps, 2),
                    algorithmic_tflop_per_step=round(tot_flop / a.steps / 1e12, 3),
                    executed_mfma_tflops=round(exe, 2), executed_frac=round(exe / peak, 4),
                    winograd_launches_per_step=len(wino) // a.steps, winograd_ms_per_step=round(sum(q[0].elapsed_time(q[1]) for q in wino) / a.steps, 2),
                    ms_per_step_with_events=round(t_events, 2),
                    note='achieved = sum over conv launches (l2i_conv2d_wino_f32 + l2i_conv2d_f32 + l2i_conv_transpose2d_f32) of 2*MAC of the dense '
                         'correlation / sum of their HIP-event durations, over a repeat of the timed steps on ONE stream with an event pair '
                         'per launch (the timed region itself runs the three loss branches on separate streams, without events); algorithmic TFLOP per image = algorithmic_tflop_per_step / batch. '
                         '3x3 stride-1 layers run as Winograd F(2x2,3x3) (fp32 products, 16/36 of the direct multiplies): executed_mfma_tflops counts what the matrix cores '
                         'actually execute, achieved counts the algorithmic FLOPs of the dense correlation as the contract asks')
    out = dict(metric='edited images/sec', value=round(value, 3), unit='images/s', n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=round(ms_per_step, 2), higher_is_better=True, scaling='weak', vs_baseline=None, dtype=a.precision,
               data='synthetic',
               config=dict(workload='StyleGAN2 FFHQ-shaped %d^2 generator (random-init), ResNet-50 regressor, %d attr, %s, '
                                    'batch %d per GPU, linear W+ walk' % (a.resolution, len(attrs),
                                                                          'reg-only loss' if a.reg_only else 'full loss (reg+content+GAN)', a.batch),
                           resolution=a.resolution, global_batch=global_b, per_gpu_batch=a.batch, attrs=attrs,
                           losses='reg' if a.reg_only else 'reg+content+gan', parallelism='dp%d' % world,
                           loss_branch_streams=3 if constants.CONCURRENT_LOSS_BRANCHES else 1,
                           loss=float(r['loss'])),
               roofline=roof, alt_precision=alt)
    if world == 1 and a.cpu_baseline_s > 0:
        out['cpu_baseline'] = cpu_baseline(a.resolution, attrs, a.cpu_baseline_s, full_loss=not a.reg_only)
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)
    dist.shutdown()


if __name__ == '__main__':
    main()
