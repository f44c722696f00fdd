"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass) of

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_f> -o run --output-format csv -- python3 bench.py --serial_streams \
        --steps 1 --warmup 1 --cpu_baseline_s 0 --no_config5 --no_reg_only --sweep none --no_kernel_events
    (same with --pmc WRITE_SIZE -d <dir_w>)

into profiles/<name>.json: HBM bytes per kernel family and per conv launch.  Corrections (MI355X_MICROARCH.md, "HBM"):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a coalesced streaming read, so it is
doubled; both factors are checked on a streaming kernel of known byte count in the same run (maxpool2x2_bwd_add_diff_kernel:
9.25 B read, 4 B written per element; an earlier build's axpby_kernel gave 2.00 and 1.00).

    python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_c3_hbm_traffic.json [c3|c5|c5x3]

Round 2: the summary also carries the bytes per launch of every conv kernel FAMILY (the unit bench.py's roofline.traffic reports for the
dominant family), the sha256 of the libl2i_hip.so that was profiled and the workload key; bench.py drops the figure when either differs.
"""
import collections
import csv
import hashlib
import json
import os
import sys

CONV = ('conv_wino_kernel', 'conv_wino4_kernel', 'conv_wino4s_kernel', 'conv_mfma_kernel', 'conv3x3s2_dma_kernel', 'gemm1x1_kernel', 'convt_mfma_kernel', 'conv_direct_small_kernel',
        'conv_bf16x3_pipe_kernel', 'conv_cin3_kernel', 'splitk_epilogue_kernel', 'conv_h8_kernel', 'conv_img_h8_kernel')
WORKLOADS = {'c3': [1024, 8, ['Smiling'], 'f32', False],
             'c5': [1024, 8, ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk'], 'f16', False],        # [r5] config 5 runs fp16 elements
             'c5bf16': [1024, 8, ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk'], 'bf16', False],
             'c5x3': [1024, 8, ['dirty', 'daylight', 'night', 'sunrisesunset', 'dawndusk'], 'bf16x3', False]}


def family_of(kernel):
    """rocprof kernel name -> the family names of latent2im_amd.conv.FAMILY_INFO."""
    if kernel.startswith('conv_wino4_kernel') or kernel.startswith('conv_wino4s_kernel'):
        return 'winograd4_f32'
    if kernel.startswith('conv_wino_kernel'):
        return 'winograd_f32'
    if kernel.startswith('conv_mfma_kernel') or kernel.startswith('conv3x3s2_dma_kernel') or kernel.startswith('splitk_epilogue'):
        return 'implicit_gemm_f32'
    if kernel.startswith('gemm1x1_kernel') or 'pair_f32_kernel' in kernel:      # [r6] the fp32 pair launch replaces two of the GEMM kernel's: same family
        return 'gemm1x1_f32'
    if kernel.startswith('convt_mfma_kernel'):
        return 'transposed_f32'
    if kernel.startswith('conv_cin3_kernel'):
        return 'cin3_f32'
    if kernel.startswith('conv_direct_small_kernel'):
        return 'direct_small_valu'
    if 'pair_h8_kernel' in kernel:               # [r6] two chained 1x1 convs of the 16-bit path in one launch: same family
        return 'conv_h8'
    if kernel.startswith('conv_img_h8_kernel'):           # [r5] the image-side convs of the 16-bit path: bench.py prices them in the conv_h8 family
        return 'conv_h8'
    if kernel.startswith('conv_h8_kernel'):               # <WM, WN, K, S, TR, OUT32, RELU_IN, KS>
        args = kernel[kernel.index('<') + 1:kernel.rindex('>')].split(',')
        return 'transposed_h8' if args[4].strip() != '0' else 'conv_h8'
    if kernel.startswith('conv_bf16x3_pipe_kernel'):
        args = kernel[kernel.index('<') + 1:kernel.rindex('>')].split(',')
        return 'transposed_bf16x3' if len(args) > 5 and args[5].strip() not in ('0',) else 'implicit_gemm_bf16x3'
    return None


def load(d, name):
    out = {}
    for r in csv.DictReader(open('%s/run_counter_collection.csv' % d)):
        if r['Counter_Name'] == name:
            kname = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            for ns in ('l2i_h8_bf16::', 'l2i_h8s_bf16::', 'l2i_h8_f16::', 'l2i_h8s_f16::', 'l2i_pair_f32::'):      # [r5] the h8 kernels live in per-element-type namespaces
                kname = kname.replace(ns, '')
            out[int(r['Dispatch_Id'])] = (kname, int(r['Grid_Size']), float(r['Counter_Value']) * 1024.0)
    return out


def main(df, dw, dst, tag='c3', steps=2, cal_elems=8 * 64 * 1024 * 1024):
    f, w = load(df, 'FETCH_SIZE'), load(dw, 'WRITE_SIZE')
    fam = collections.defaultdict(lambda: dict(launches=0, fetch_raw=0.0, write_raw=0.0))
    for i, (k, g, v) in f.items():
        a = fam[k]
        a['launches'] += 1
        a['fetch_raw'] += v
    for i, (k, g, v) in w.items():
        fam[k]['write_raw'] += v
    # calibration on a kernel of known byte count in the same run: maxpool2x2_bwd_add_diff_kernel runs once per step on the VGG conv1_2 tap
    # [8, 64, 1024, 1024] of the default workload: per output element 8 B read (a, b) + 5/4 B (pooled gradient, arg-max byte), 4 B written
    cal_r = cal_w = None
    ax = [(g, v, w[i][2]) for i, (k, g, v) in f.items() if k == 'maxpool2x2_bwd_add_diff_kernel' and i in w and w[i][0] == k]
    if ax:
        n = cal_elems * len(ax)
        cal_r = (8.0 + 1.25) * n / sum(v for _, v, _ in ax)
        cal_w = 4.0 * n / sum(x for _, _, x in ax)
    fr, fw = 2.0, 1.0
    rows = {}
    conv = dict(launches=0, bytes=0.0)
    for k, a in sorted(fam.items(), key=lambda kv: -(fr * kv[1]['fetch_raw'] + fw * kv[1]['write_raw'])):
        b = fr * a['fetch_raw'] + fw * a['write_raw']
        rows[k] = dict(launches_per_step=a['launches'] / steps, read_GB_per_step=round(fr * a['fetch_raw'] / steps / 1e9, 3),
                       write_GB_per_step=round(fw * a['write_raw'] / steps / 1e9, 3))
        if any(k.startswith(c) for c in CONV):
            conv['launches'] += 0 if k.startswith('splitk_epilogue') else a['launches']      # the second pass of a split-K call is not a call
            conv['bytes'] += b
    per_family = collections.defaultdict(lambda: dict(launches=0, bytes=0.0))
    for k, a in fam.items():
        f_ = family_of(k)
        if f_:
            per_family[f_]['launches'] += 0 if k.startswith('splitk_epilogue') else a['launches']
            per_family[f_]['bytes'] += fr * a['fetch_raw'] + fw * a['write_raw']
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'latent2im_amd', 'libl2i_hip.so')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from latent2im_amd import _lib as l2i_lib
    out = dict(lib_sha256_16=hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16], src_sha256_16=l2i_lib.source_hash(), workload=WORKLOADS[tag],
               per_family={k: dict(launches_per_step=v['launches'] / steps, bytes_per_launch=v['bytes'] / max(v['launches'], 1),
                                   GB_per_step=round(v['bytes'] / steps / 1e9, 3)) for k, v in per_family.items()},
               source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --serial_streams, %d steps' % steps,
               corrections=dict(unit='KiB -> bytes', fetch_factor=fr, write_factor=fw,
                                calibration=dict(kernel='maxpool2x2_bwd_add_diff_kernel', read_factor_measured=cal_r, write_factor_measured=cal_w,
                                                       note='known bytes / reported bytes on that kernel in the same run')),
               conv_launches_per_step=conv['launches'] / steps, conv_bytes_per_step=conv['bytes'] / steps,
               conv_bytes_per_launch=conv['bytes'] / max(conv['launches'], 1),
               all_kernels_GB_per_step=round(sum(fr * a['fetch_raw'] + fw * a['write_raw'] for a in fam.values()) / steps / 1e9, 2),
               per_kernel=rows)
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('corrections', 'conv_launches_per_step', 'conv_bytes_per_step', 'conv_bytes_per_launch', 'all_kernels_GB_per_step')}, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:5])
