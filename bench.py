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

The line carries `step_ms` (one HIP event per timed step on the main stream: per-step times, median / min / max), `sensors`
(shader clock and socket power before / during / after the timed region), `config5` (the c5 workload on the 16-bit path, hipGraph, with its
own roofline) and `reg_only` (--no_content_loss --no_gan_loss: the one loss configuration SURVEY 8d calls jointly reachable) beside the headline.

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
C5_PRECISION = 'f16'                # [r5] config 5's 16-bit path: IEEE fp16 h8 maps (BASELINE configs[4]: "fp16 MFMA"); 'bf16' = rounds 3-4 (--precision bf16)


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
    # `cores` = what the port really ran on: the CPUs the host gives this process (cgroup quota / affinity: the GPU box shows 256 hardware threads
    # to a pod with a quota of 16 CPUs; more runnable threads than that are only throttled — 16 threads 5.3 s, 32 threads 7.1 s, 256 threads 334 s
    # for the same step), at most 32
    import oracle
    threads = min(oracle.host_cpus(), 32)
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
    out = dict(value=done / t_total, unit='images/s', cores=threads, kind='port', cpu=cpu_model(), host_threads=os.cpu_count(), usable_cpus=oracle.host_cpus(),
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


def source_hash():
    from latent2im_amd import _lib
    return _lib.source_hash()


def committed_build():
    """latent2im_amd/csrc/BUILD_HASHES.json (tools/update_build_hash.py): the library sha256 the committed sources build to (reproducible build),
    and whether the library this run loaded is that build."""
    from latent2im_amd import _lib
    rec = _lib.committed_build()
    if rec is None:
        return None
    return dict(library_sha256_16=rec['library_sha256'][:16], kernel_sources_sha256_16=rec['kernel_sources_sha256_16'],
                sources_match=rec['sources_match'], library_match=rec['library_match'])


def _drm_card_of(index):
    """/sys/class/drm/cardN/device of HIP device `index`, matched by PCI address (a box can expose dozens of cards: card0 is NOT device 0)."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(index)
        want = '%04x:%02x:%02x.' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except (AttributeError, RuntimeError, AssertionError):
        return None
    for d in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        if os.path.basename(os.path.realpath(d)).startswith(want) and os.path.isfile(os.path.join(d, 'pp_dpm_sclk')):
            return d
    return None


def gpu_sensors(index=0):
    """{sclk_mhz, power_w, source}: current shader clock and socket power of HIP device `index`, from amdgpu's sysfs files of the card with the
    device's PCI address when readable (microseconds, no subprocess: usable between steps), else one `rocm-smi --showclocks --showpower` call;
    fields are None when neither answers (never another card's numbers)."""
    import glob
    out = dict(sclk_mhz=None, power_w=None, source=None)
    d = gpu_sensors.__dict__.setdefault('card', {}).get(index, False)
    if d is False:
        d = gpu_sensors.card[index] = _drm_card_of(index)
    if d:
        try:
            for line in open(os.path.join(d, 'pp_dpm_sclk')):
                if '*' in line:
                    out['sclk_mhz'] = float(line.split(':')[1].lower().replace('mhz', '').replace('*', '').strip())
            for pat in ('hwmon/hwmon*/power1_input', 'hwmon/hwmon*/power1_average'):
                for f in glob.glob(os.path.join(d, pat)):
                    out['power_w'] = float(open(f).read().strip()) / 1e6
                    break
                if out['power_w'] is not None:
                    break
            out['source'] = 'sysfs:%s (%s)' % (d.split('/')[4], os.path.basename(os.path.realpath(d)))
        except (OSError, ValueError, IndexError):
            pass
    if out['sclk_mhz'] is None and out['power_w'] is None and not os.environ.get('L2I_NO_ROCM_SMI'):
        import re
        import subprocess
        try:
            txt = subprocess.run(['rocm-smi', '-d', str(index), '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r'sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)', txt)
            if m:
                out['sclk_mhz'] = float(m.group(1))
            m = re.search(r'Power \(W\):\s*([0-9.]+)', txt)
            if m:
                out['power_w'] = float(m.group(1))
            out['source'] = 'rocm-smi'
        except (OSError, subprocess.SubprocessError):
            pass
    return out


def timed_steps(one_step, first, steps, max_ahead, sensor_index=None, sensor_every=5):
    """EXACTLY `steps` steps bracketed by barrier + device synchronisation on both sides (wall clock = the reported time), with one HIP
    event recorded on the main stream after every step (the step's last launches — all-reduce, Adam — are on it): per-step GPU times
    without any host synchronisation of the newest `max_ahead` steps.  The host waits for step i - max_ahead before launching step i
    (the reference synchronises every step, train.py:110; unbounded run-ahead only grows the caching allocator's footprint — blocks used on
    the loss-branch streams are handed back when their events complete).  Returns (wall seconds, [ms per step], last result, sensors)."""
    from latent2im_amd import dist
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    sens = []
    cheap = sensor_index is not None and gpu_sensors.__dict__.get('cheap', False)
    t0 = time.perf_counter()
    evs[0].record()
    r = None
    for i in range(steps):
        if max_ahead > 0 and i >= max_ahead:
            evs[i - max_ahead + 1].synchronize()
        r = one_step(first + i)
        evs[i + 1].record()
        if cheap and i % sensor_every == sensor_every - 1:
            sens.append(dict(gpu_sensors(sensor_index), after_launching_step=i + 1))
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)], r, sens


def step_stats(ms):
    srt = sorted(ms)
    n = len(srt)
    return dict(step_ms=[round(x, 2) for x in ms], median_ms=round(srt[n // 2] if n % 2 else 0.5 * (srt[n // 2 - 1] + srt[n // 2]), 2),
                min_ms=round(srt[0], 2), max_ms=round(srt[-1], 2))


def call_bytes(q):
    """Algorithmic HBM bytes of a conv call: input (+ its gradient mask) read once, output written once, every fused epilogue
    operand (residual, its mask, output mask, accumulate, res_sub) read once; weights / per-sample vectors are noise."""
    B, cin, cout, kh, kw, stride, H, W, OH, OW, step, in_mask = q[3][:12]
    if q[4] == 'l2i_conv1x1_pair_f32':                      # [r6] the fp32 pair: input, identity, the wide map written, the narrow output
        return 4.0 * B * H * W * (cin + 2 * cout + q[3][14])
    if q[4].startswith(('l2i_conv1x1_pair_h8', 'l2i_conv_chain3_h8')):                                 # [r6] two chained 1x1 convs: input, the first conv's operand map, the wide map written, the narrow output
        return 2.0 * B * H * W * (cin + 2 * cout + q[3][14])            # (chain3: cin = the 3x3 conv's input, read once plus its halo)
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


def roofline_block(prof, steps, tag, workload_key, t_events, ev_steps=None):
    """`roofline` object from the per-launch event pass (`prof`): dominant conv family, the roof it is nearer to, every family, the PMC
    traffic figure of profiles/ when it was taken on these kernel sources and this workload."""
    from latent2im_amd import conv
    fams = family_table(prof, steps)
    dom = fams[0]                                           # the family with the most GPU time per step
    tot_ms = sum(q[0].elapsed_time(q[1]) for q in prof)
    tot_flop = sum(q[2] for q in prof)
    exe_flop = sum(q[2] * conv.FAMILY_INFO[q[5]][1] for q in prof)
    peak_all = max(f['peak_tflops'] for f in fams)
    traffic, traffic_note = None, 'no PMC summary under profiles/ for these kernel sources and this workload'
    src = source_hash()
    for rnd in ('r06', 'r05', 'r04', 'r03'):
        tp = os.path.join(ROOT, 'profiles', '%s_%s_hbm_traffic.json' % (rnd, tag))
        if not os.path.isfile(tp):
            continue
        tj = json.load(open(tp))
        ent = tj.get('per_family', {}).get(dom['family'])
        same_build = tj.get('src_sha256_16') == src or ('src_sha256_16' not in tj and tj.get('lib_sha256_16') == lib_hash())
        if same_build and tj.get('workload') == workload_key and ent and abs(ent['launches_per_step'] - dom['launches_per_step']) < 0.5:
            traffic = round(ent['bytes_per_launch'])
            traffic_note = ('HBM bytes per launch of %s from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) over this command with '
                            '--serial_streams (profiles/%s, tools/hbm_traffic.py): KiB -> bytes, FETCH_SIZE doubled (gfx950), both factors checked on '
                            'a kernel of known byte count in the same run; the summary carries the sha256 of the kernel SOURCES it was taken with '
                            '(%s) and is dropped when they differ' % (dom['kernel'], os.path.basename(tp), src))
            break
        traffic_note = 'profiles/%s was taken with other kernel sources / workload (sources %s now): dropped' % (os.path.basename(tp), src)
    hbm_frac = round(dom['hbm_GBs_algorithmic'] / HBM_PEAK_GBS, 4)
    if hbm_frac > dom['frac']:                              # the 16-bit kernels: closer to the HBM roof than to the bf16 matrix roof
        head = dict(bound='hbm', kernel=dom['kernel'], family=dom['family'], achieved=dom['hbm_GBs_algorithmic'], peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=hbm_frac, mfma_bound=dict(peak_TFLOPs=dom['peak_tflops'], achieved_TFLOPs_executed=dom['executed_tflops'], frac=dom['frac']))
    else:
        head = dict(bound='mfma', kernel=dom['kernel'], family=dom['family'], achieved=dom['executed_tflops'], peak=dom['peak_tflops'],
                    unit='TFLOP/s', frac=dom['frac'])
    return dict(head,
                traffic=traffic, traffic_unit='bytes per launch', traffic_note=traffic_note,
                algorithmic_tflops=dom['algorithmic_tflops'], algorithmic_flop_per_launch=dom['algorithmic_flop_per_launch'],
                algorithmic_bytes_per_launch=dom['algorithmic_bytes_per_launch'], avg_launch_ms=dom['avg_launch_ms'],
                launches_per_step=dom['launches_per_step'], kernel_ms_per_step=dom['ms_per_step'],
                hbm_bound=dict(peak_GBs=HBM_PEAK_GBS, achieved_GBs_algorithmic=dom['hbm_GBs_algorithmic'], frac=hbm_frac),
                families=fams,
                all_conv=dict(launches_per_step=len(prof) // steps, ms_per_step=round(tot_ms / steps, 2),
                              algorithmic_tflop_per_step=round(tot_flop / steps / 1e12, 3),
                              algorithmic_tflops=round(tot_flop / (tot_ms * 1e-3) / 1e12, 2),
                              executed_tflops=round(exe_flop / (tot_ms * 1e-3) / 1e12, 2),
                              executed_frac_of_fastest_instruction_peak=round(exe_flop / (tot_ms * 1e-3) / 1e12 / peak_all, 4)),
                ms_per_step_with_events=round(t_events, 2), event_pass_step_ms=ev_steps, library_sha256_16=lib_hash(), kernel_sources_sha256_16=src, committed_build=committed_build(),
                note='dominant kernel family = most GPU time per step; bound = the roof it is closer to (hbm: algorithmic bytes of its launches '
                     '/ their HIP-event time against 8 TB/s).  mfma: achieved / frac = FLOPs the matrix cores EXECUTE in that family '
                     '(Winograd F(2x2,3x3): 16/36, F(4x4,3x3): 36/144 of the dense correlation; 3-term bf16 split: 3 products per fp32 product) '
                     '/ its HIP-event time / the dense peak of the MFMA instruction it runs on; algorithmic_tflops = 2*MAC of the dense '
                     'correlation / the same time. Events: an eager repeat of the timed steps on ONE stream with an event pair per conv launch '
                     '(the timed region has none). families: the same figures for every conv kernel family of the step; all_conv: their sum')


class Workload:
    """One benchmarked configuration: the graph on synthetic weights, its z stream and alpha draws, eager or hipGraph-replayed steps."""

    def __init__(self, cfg, precision, resolution, batch, attrs, world, n_steps, use_graph):
        from latent2im_amd import conv, selfcheck, synth, dist
        self.c5 = cfg == 'c5'
        self.precision, self.resolution, self.batch, self.world = precision, resolution, batch, world
        self.attrs = attrs or (SCENE_ATTRS if self.c5 else ['Smiling'])
        self.clamp = self.c5                                 # train_multi_attr.py:113 flow for the multi-attribute configs
        self.use_graph = use_graph
        conv.PRECISION = precision
        np.random.seed(1234)
        self.g = selfcheck.build_graph(resolution, self.attrs, batch * world, lr=1e-4, transform='scene' if self.c5 else 'face')
        if world > 1:
            dist.broadcast_parameters(self.g.walk.parameters())
        self.global_b = batch * world
        self.zs_all = synth.z_sample(self.global_b * n_steps, seed=0)
        self.n_steps = n_steps
        self.sl = dist.shard(self.global_b)
        self.captured = {}

    def key(self, reg_only=False):
        return [self.resolution, self.batch, self.attrs, self.precision, bool(reg_only)]

    def draw_alpha(self, batch=None):                        # Face/SceneTransform.get_train_alpha: one draw per step, shared by the batch
        lo = -1.0 if self.c5 else 0.0
        return np.ones((batch or self.batch, len(self.attrs))) * np.random.uniform(lo, 1, len(self.attrs))

    def stepper(self, reg_only=False, graph=None):
        """i -> one training step on batch i of the z stream (wrapping around), eager or replayed from the hipGraph captured for these flags."""
        from latent2im_amd import capture, conv, selfcheck
        conv.PRECISION = self.precision
        flags = dict(no_content_loss=reg_only, no_gan_loss=reg_only)
        graph = self.use_graph if graph is None else graph
        cap = None
        if graph:
            if reg_only not in self.captured:
                self.captured[reg_only] = capture.CapturedStep(self.g, self.batch, len(self.attrs), clamp=self.clamp, **flags)
            cap = self.captured[reg_only]

        def one_step(i):
            i %= self.n_steps
            zs = self.zs_all[i * self.global_b:(i + 1) * self.global_b][self.sl]
            if cap is not None:
                return cap(zs, self.draw_alpha())
            return selfcheck.run_step(self.g, zs, self.draw_alpha(), clamp=self.clamp, **flags)
        return one_step

    def release(self):
        import gc
        self.captured = {}
        self.g = None
        gc.unfreeze()                                        # (warm_up froze this graph's objects: a cycle among them must stay collectable)
        gc.collect()
        torch.cuda.empty_cache()

    def describe(self, reg_only, noise, graph):
        return ('StyleGAN2 FFHQ-shaped %d^2 generator (random-init, noise weights ~ %g: per-layer noise drawn every pass), ResNet-50 regressor, '
                '%d %s attr (%s flow), %s, batch %d per GPU, linear W+ walk, %s launches'
                % (self.resolution, noise, len(self.attrs), 'transient-scene' if self.c5 else 'CelebA',
                   'train_multi_attr.py clamp' if self.clamp else 'train.py', 'reg-only loss' if reg_only else 'full loss (reg+content+GAN)',
                   self.batch, 'hipGraph-replayed' if graph else 'eager'))


def warm_up(one_step, count, seconds, dev=None):
    """`count` untimed steps, then more until about `seconds` of wall time have passed (clocks, power state, allocator pools, lazily packed
    weights settle on a time scale, not a step count).  Every rank runs the SAME number of steps (a step holds the walk-gradient all-reduce: a
    per-rank time test would leave ranks in different collectives): the extra steps are computed once from the slowest rank's time for the
    counted ones.  Returns the number of steps run."""
    import math
    from latent2im_amd import capture, dist
    t0 = time.perf_counter()
    n = 0
    for _ in range(max(count, 1)):
        one_step(n)
        n += 1
        if n == 1:
            capture.freeze_host_objects()                  # as trainer.train does after its first step
    torch.cuda.synchronize()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0, dev)
    if elapsed < seconds:
        per_step = max(elapsed / n if n > 1 else elapsed, 1e-3)    # (the very first step carries the lazy weight packing)
        extra = min(int(math.ceil((seconds - elapsed) / per_step)), 200)
        for _ in range(extra):
            one_step(n)
            n += 1
        torch.cuda.synchronize()
    return n


def event_pass(wl, one_step, first, steps):
    """The same steps again, eager on ONE stream, with a HIP event pair around every conv launch on the launch stream (kept out of the timed
    region: ~280 pairs per step cost ~7 % wall on their own, and concurrent streams make per-kernel durations overlap).
    [r5] One UNTIMED step goes first and its entries are dropped: the pass creates ~550 events per step, and the first step pays for their
    creation (hipEventCreate through torch's event pool, empty until then) — round 4's driver record showed `ms_per_step_with_events` at 205-216 ms
    in some runs and 116 in others with identical per-kernel times; the per-step wall times (`event_pass_step_ms`, one synchronisation per step)
    now say where such a stall sits.  Returns (per-launch entries of the `steps` counted steps, mean wall ms per counted step, [wall ms per step
    INCLUDING the dropped first one])."""
    from latent2im_amd import constants, conv
    concurrent, constants.CONCURRENT_LOSS_BRANCHES = constants.CONCURRENT_LOSS_BRANCHES, False
    torch.cuda.synchronize()
    per_step, prof = [], []
    for i in range(steps + 1):
        conv.PROFILE = []
        t1 = time.perf_counter()
        one_step(first + (i - 1 if i else 0))
        torch.cuda.synchronize()
        per_step.append(round((time.perf_counter() - t1) * 1e3, 2))
        if i:
            prof += conv.PROFILE
    conv.PROFILE = None
    constants.CONCURRENT_LOSS_BRANCHES = concurrent
    return prof, sum(per_step[1:]) / steps, per_step


def quick_rate(one_step, steps, global_b, dev, max_ahead):
    """images/s of `steps` steps after one untimed one (the secondary figures of the line: reg-only, config 5's eager twin)."""
    from latent2im_amd import dist
    one_step(0)
    el, ms, _, _ = timed_steps(one_step, 1, steps, max_ahead)
    el = dist.max_over_ranks(el, dev)
    return dict(value=round(global_b * steps / el, 3), unit='images/s', ms_per_step=round(el / steps * 1e3, 2), steps=steps, **step_stats(ms))


def one_rank_allreduce_us(wgrad):
    """N = 1 has no process group; the step's only collective is still rehearsed: a ONE-rank RCCL group is built after all timing, the
    walk-gradient-shaped all-reduce is timed through it, and the group is destroyed.  (value or None, note)."""
    from latent2im_amd import dist
    import socket
    try:
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ['L2I_FORCE_PG'] = '1'
        dist.init_from_env()
        us = dist.time_allreduce(wgrad.detach())
        backend = torch.distributed.get_backend()
        dist.shutdown()
        return us, ('ONE-rank %s (RCCL) group built after the timed regions as a rehearsal of the step\'s only collective: median of 100 synchronised '
                    'all-reduces of a walk-gradient-shaped tensor (%d bytes); says nothing about xGMI — the N > 1 figure is measured over the real group'
                    % (backend, wgrad.numel() * 4))
    except Exception as e:                                   # a box without a working RCCL must not lose the bench line
        return None, 'one-rank RCCL rehearsal failed: %s: %s' % (type(e).__name__, str(e)[:200])
    finally:
        os.environ.pop('L2I_FORCE_PG', None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--warmup_s', type=float, default=3.0, help='keep warming up (beyond --warmup steps) until this many seconds have passed')
    ap.add_argument('--max_ahead', type=int, default=2, help='host launches at most this many steps ahead of the GPU (0: unbounded)')
    ap.add_argument('--no_sensors', action='store_true', help='no clock / power reads at all (diagnostic: do the reads themselves disturb the GPU?)')
    ap.add_argument('--no_gc', action='store_true', help='disable the Python garbage collector during the timed region (diagnostic)')
    ap.add_argument('--config', default='c3', choices=['c3', 'c5'], help='BASELINE.json configs[2] (headline) or configs[4] per-GPU shape')
    ap.add_argument('--resolution', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch')
    ap.add_argument('--attrs', type=str, default=None)
    ap.add_argument('--reg_only', action='store_true', help='--no_content_loss --no_gan_loss as the headline workload')
    ap.add_argument('--cpu_baseline_s', type=float, default=12.0, help='CPU-oracle time budget (0 = skip)')
    ap.add_argument('--no_kernel_events', action='store_true', help='do not bracket conv launches with events')
    ap.add_argument('--event_steps', type=int, default=5, help='steps of the per-launch event pass (roofline)')
    ap.add_argument('--precision', default=None, choices=['f32', 'bf16x3', 'bf16', 'f16'],
                    help="f32: exact fp32 MFMA, fp32 storage (c3 default); f16: the 16-bit path with IEEE fp16 h8 storage, one fp16 MFMA per MAC, fp32 accumulation, "
                         "scaled gradients with a device-side overflow guard (c5 default since round 5); bf16: the same path with bfloat16 elements (c5 default of "
                         "rounds 3-4: profiles/r03_*, r04_* `config5` / `reg_only.bf16` keys are bf16 numbers); bf16x3: fp32 storage, 3-term bf16 split on the "
                         "matrix cores (fp32-class results)")
    ap.add_argument('--hip_graph', type=int, default=None, help='1: replay forward+backward from one hipGraph (c5 default), 0: eager launches')
    ap.add_argument('--no_config5', action='store_true', help='skip the `config5` block (c5 workload, 16-bit path, hipGraph) of a c3 run')
    ap.add_argument('--config5_steps', type=int, default=20, help='timed steps of the config5 block (20: one host hiccup of 20 ms moves the figure by 3 %, not 6)')
    ap.add_argument('--no_reg_only', action='store_true', help='skip the `reg_only` figures')
    ap.add_argument('--no_allreduce_rehearsal', action='store_true', help='N = 1: do not build the one-rank RCCL group for `allreduce_us` (profiler runs)')
    ap.add_argument('--direct_3x3', action='store_true', help='run 3x3 stride-1 layers on the direct implicit-GEMM kernel instead of Winograd')
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
    t_start = time.perf_counter()
    c5 = a.config == 'c5'
    precision = a.precision or (C5_PRECISION if c5 else 'f32')
    use_graph = bool(a.hip_graph) if a.hip_graph is not None else c5
    if a.serial_streams:
        constants.CONCURRENT_LOSS_BRANCHES = False
    constants.SYNTH_NOISE_STRENGTH = a.noise_strength
    conv.USE_WINOGRAD = not a.direct_3x3
    rk, world, local = dist.init_from_env()
    if a.gpus != world:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    ranks = dist.assert_distinct_devices()          # N ranks over RCCL = N distinct physical GPUs (host + PCI address), checked before any work; reported as ranks_seen
    assert torch.cuda.is_available(), 'bench.py needs the MI355X'
    dev = torch.device('cuda', torch.cuda.current_device())
    sens_idx = local
    none = dict(sclk_mhz=None, power_w=None, source='off (--no_sensors)')
    s0 = none if a.no_sensors else gpu_sensors(sens_idx)
    gpu_sensors.cheap = bool(s0['source'] and s0['source'].startswith('sysfs'))
    wl = Workload(a.config, precision, a.resolution, a.batch, a.attrs.split(',') if a.attrs else None, world, a.steps + a.warmup, use_graph)
    global_b = wl.global_b
    one_step = wl.stepper(reg_only=a.reg_only)

    # ---- the headline: W (+ time-based) warm-up steps, then EXACTLY K timed steps
    warm_run = warm_up(one_step, a.warmup, a.warmup_s, dev)
    mem0 = torch.cuda.memory_stats()
    sens_before = none if a.no_sensors else gpu_sensors(sens_idx)       # (the device idles for the synchronisation in front of it: expect idle clocks)
    if a.no_gc:
        import gc
        gc.collect()
        gc.disable()
    elapsed, step_ms, r, sens_mid = timed_steps(one_step, a.warmup, a.steps, a.max_ahead, None if a.no_sensors else sens_idx)
    if a.no_gc:
        gc.enable()
    sens_after = none if a.no_sensors else gpu_sensors(sens_idx)
    mem1 = torch.cuda.memory_stats()
    per_rank_ms = [round(t / a.steps * 1e3, 3) for t in dist.gather_floats(elapsed)]
    elapsed = dist.max_over_ranks(elapsed, dev)
    loss_value = float(r['loss'])
    wgrad = next(iter(wl.g.walk.parameters()))
    allreduce_us = dist.time_allreduce(wgrad.detach())
    allreduce_note = ('median of 100 synchronised all-reduces of a walk-gradient-shaped tensor (%d bytes) over the %s process group'
                      % (wgrad.numel() * 4, torch.distributed.get_backend())) if allreduce_us is not None else None
    alloc = dict(reserved_GB_before=round(mem0['reserved_bytes.all.current'] / 1e9, 2), reserved_GB_after=round(mem1['reserved_bytes.all.current'] / 1e9, 2),
                 device_mallocs_in_timed_region=int(mem1['segment.all.allocated'] - mem0['segment.all.allocated']),
                 alloc_retries=int(mem1['num_alloc_retries'] - mem0['num_alloc_retries']))

    # ---- roofline of the conv kernels (every rank repeats the steps: the walk-gradient all-reduce is inside a step; rank 0 reports)
    prof, t_events, ev_steps = None, None, None
    n_ev = min(a.event_steps, a.steps)
    if not a.no_kernel_events:
        prof, t_events, ev_steps = event_pass(wl, wl.stepper(reg_only=a.reg_only, graph=False), a.warmup, n_ev)
    dist.barrier()

    # ---- reg-only loss (train.py:174-176 --no_content_loss --no_gan_loss) on the same graph
    reg = None
    if not a.no_reg_only and not a.reg_only:
        reg = {precision: dict(quick_rate(wl.stepper(reg_only=True), 5, global_b, dev, a.max_ahead),
                               workload=wl.describe(True, a.noise_strength, use_graph))}
        if not a.no_kernel_events:                           # [r5] the reg-only loss is the configuration SURVEY 8(d) calls jointly reachable: its own roofline
            prof_r, t_ev_r, ev_r = event_pass(wl, wl.stepper(reg_only=True, graph=False), 0, n_ev)
            reg[precision]['roofline'] = roofline_block(prof_r, n_ev, 'c3_reg', wl.key(True), t_ev_r, ev_r) if rk == 0 else None

    # ---- batch sweep (the BASELINE metric is quoted over a batch sweep): the same eager step at other per-GPU batches, every rank
    sweep = None
    if a.sweep and a.sweep not in ('0', 'none', 'off'):
        import gc
        wl.captured = {}
        gc.collect()
        torch.cuda.empty_cache()
        sweep = []
        flags = dict(no_content_loss=a.reg_only, no_gan_loss=a.reg_only)
        for bs in [int(x) for x in a.sweep.split(',') if x]:
            gb = bs * world
            zs_s = synth.z_sample(gb * (a.sweep_steps + 1), seed=1)
            sl_s = dist.shard(gb)

            def sweep_step(i):
                zs = zs_s[i * gb:(i + 1) * gb][sl_s]
                return selfcheck.run_step(wl.g, zs, wl.draw_alpha(bs), clamp=wl.clamp, **flags)
            try:
                torch.cuda.reset_peak_memory_stats()
                sweep_step(0)
                el, ms, _, _ = timed_steps(sweep_step, 1, a.sweep_steps, a.max_ahead)
                t_s = dist.max_over_ranks(el, dev)
                sweep.append(dict(batch=bs, global_batch=gb, images_s=round(gb * a.sweep_steps / t_s, 3), ms_per_step=round(t_s / a.sweep_steps * 1e3, 2),
                                  min_ms=round(min(ms), 2), peak_mem_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1)))
            except torch.OutOfMemoryError as e:
                sweep.append(dict(batch=bs, global_batch=gb, images_s=None, error='out of memory: %s' % str(e)[:120]))
            gc.collect()
            torch.cuda.empty_cache()

    # ---- config 5 (BASELINE configs[4] per-GPU shape) beside the headline: scene attributes, clamp flow, 16-bit path, hipGraph-replayed
    cfg5 = None
    if not c5 and not a.no_config5 and a.resolution == 1024:
        wkey3, desc3 = wl.key(a.reg_only), wl.describe(a.reg_only, a.noise_strength, use_graph)
        wl.release()
        w5 = Workload('c5', C5_PRECISION, a.resolution, a.batch, None, world, a.config5_steps + 2, True)
        step5 = w5.stepper()
        warm5 = warm_up(step5, 2, 1.0, dev)
        el5, ms5, r5, _ = timed_steps(step5, 2, a.config5_steps, a.max_ahead)
        el5 = dist.max_over_ranks(el5, dev)
        roof5 = None
        if not a.no_kernel_events:
            prof5, t_ev5, ev5 = event_pass(w5, w5.stepper(graph=False), 2, min(a.event_steps, a.config5_steps))
            roof5 = roofline_block(prof5, min(a.event_steps, a.config5_steps), 'c5', w5.key(), t_ev5, ev5) if rk == 0 else None
        cfg5 = dict(value=round(global_b * a.config5_steps / el5, 3), unit='images/s', ms_per_step=round(el5 / a.config5_steps * 1e3, 2),
                    steps=a.config5_steps, warmup_steps_run=warm5, dtype=C5_PRECISION, baseline_config='configs[4] per-GPU shape',
                    workload=w5.describe(False, a.noise_strength, True), loss=float(r5['loss']), roofline=roof5,
                    # fp16: the dynamic loss scale after every step this workload ran (warm-up, timed, event pass): `skipped` = optimiser steps the device-side
                    # overflow guard dropped (read after the timed regions: the guard itself never synchronises)
                    loss_scale=(dict(w5.g.loss_scaler.stats(), static_log2=dict(w5.g.loss_scaler.log2)) if getattr(w5.g, 'loss_scaler', None) is not None else None),
                    **step_stats(ms5))
        if reg is not None:
            reg[C5_PRECISION] = dict(quick_rate(w5.stepper(reg_only=True), 5, global_b, dev, a.max_ahead), workload=w5.describe(True, a.noise_strength, True))
            if not a.no_kernel_events:
                prof_r, t_ev_r, ev_r = event_pass(w5, w5.stepper(reg_only=True, graph=False), 0, n_ev)
                reg[C5_PRECISION]['roofline'] = roofline_block(prof_r, n_ev, 'c5_reg', w5.key(True), t_ev_r, ev_r) if rk == 0 else None
        w5.release()
        conv.PRECISION = precision
    else:
        wkey3, desc3 = wl.key(a.reg_only), wl.describe(a.reg_only, a.noise_strength, use_graph)

    if world == 1 and allreduce_us is None and not a.no_allreduce_rehearsal:
        allreduce_us, allreduce_note = one_rank_allreduce_us(wgrad)
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
        rows = [dict(family=k[0], shape=list(k[1:]), launches_per_step=v[0] / n_ev, ms_per_launch=round(v[1] / v[0], 4), ms_per_step=round(v[1] / n_ev, 3),
                     tflops=round(v[2] / (v[1] / v[0] * 1e-3) / 1e12, 1), GBs=round(v[3] / (v[1] / v[0] * 1e-3) / 1e9, 1)) for k, v in agg.items()]
        rows.sort(key=lambda r: -r['ms_per_step'])
        json.dump(rows, open(a.dump_launches, 'w'), indent=0)
    if prof:
        roof = roofline_block(prof, n_ev, 'c5' if c5 else 'c3', wkey3, t_events, ev_steps)
    ls_main = dict(wl.g.loss_scaler.stats(), static_log2=dict(wl.g.loss_scaler.log2)) if getattr(wl.g, 'loss_scaler', None) is not None else None
    out = dict(metric='edited images/sec', value=round(value, 3), unit='images/s', n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=round(ms_per_step, 2), higher_is_better=True, scaling='weak', vs_baseline=None, dtype=precision,
               data='synthetic',
               config=dict(workload=desc3, baseline_config='configs[4] per-GPU shape' if c5 else 'configs[2]',
                           resolution=a.resolution, global_batch=global_b, per_gpu_batch=a.batch, attrs=wkey3[2],
                           losses='reg' if a.reg_only else 'reg+content+gan', parallelism='dp%d' % world,
                           loss_branch_streams=3 if constants.CONCURRENT_LOSS_BRANCHES else 1, hip_graph=use_graph,
                           noise_strength=a.noise_strength, loss=loss_value),
               warmup_steps_run=warm_run, max_steps_ahead=a.max_ahead, **step_stats(step_ms),
               sensors=dict(at_start=s0, before_timed=sens_before, during_timed=sens_mid, after_timed=sens_after,
                            note='shader clock (MHz) / socket power (W) of this rank\'s GPU; during_timed only when the sysfs files are readable '
                                 '(no subprocess between steps); the host is up to max_steps_ahead steps ahead of the GPU when it samples'),
               allocator=alloc,
               roofline=roof, config5=cfg5, reg_only=reg, batch_sweep=sweep, loss_scale=ls_main,
               ranks_seen=ranks, per_rank_ms_per_step=per_rank_ms, allreduce_us=None if allreduce_us is None else round(allreduce_us, 1),
               allreduce_note=allreduce_note)
    if world == 1 and a.cpu_baseline_s > 0:
        out['cpu_baseline'] = cpu_baseline(a.resolution, len(wkey3[2]), a.cpu_baseline_s, full_loss=not a.reg_only, clamp=c5)
    else:
        out['cpu_baseline'] = None
    out['bench_wall_s'] = round(time.perf_counter() - t_start, 1)
    print(json.dumps(out), flush=True)
    dist.shutdown()


if __name__ == '__main__':
    main()
