"""Per-shape time table (fwd / dgrad / wgrad) of the last step in a rocprofv3 kernel trace of bench.py."""
import csv, glob, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd.yolo.nets.engine import arch
path = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem_kernel<0>' in r['Kernel_Name']]      # first convolution kernel of a step (stem statistics pass)
last = rows[idx[-1]:]
specs = arch()[1:]      # the stem runs in csrc/stem_kernels.hip (no igemm / wgrad_kernel launch): listed separately below
def hw(s):
    n = s.name
    if n == 'backbone.conv1': return 640
    if n.startswith('backbone.layer'): return 640 >> int(n[len('backbone.layer')])
    if n.startswith('embedding0') or n.startswith('embedding1_cbl'): return 20
    if n.startswith('embedding1') or n.startswith('embedding2_cbl'): return 40
    return 80
def dur(r): return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
fw = [r for r in last if 'igemm' in r['Kernel_Name'] and (', 0, ' in r['Kernel_Name'].split('igemm')[1][:60] or True)]
# forward igemm launches = first 74 igemm-family kernels of the step (before any wgrad)
first_w = next(i for i, r in enumerate(last) if 'wgrad_kernel' in r['Kernel_Name'])
# fused path (csrc/stem_l1_kernels.hip): backbone.layer1.ds_conv's forward runs inside stem_l1_kernel (with the stem activation), 73 igemm launches remain
fused = [r for r in last if 'stem_l1_kernel' in r['Kernel_Name']]
NF = 73 if fused else 74
fwd = fused[:1] + sorted([r for r in last if 'igemm' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))[:NF]
bwd = last[first_w - 10:]
wg = sorted([r for r in last if 'wgrad_kernel' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
for s, r in zip(specs, fwd):
    k = (s.cin, s.cout, s.k, s.stride, hw(s)); a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += dur(r)
    kk = s.cin * s.k * s.k
    a[4] += 2.0 * 32 * hw(s) ** 2 * s.cout * kk
for s, r in zip(reversed(specs), wg):
    agg[(s.cin, s.cout, s.k, s.stride, hw(s))][3] += dur(r)
# dgrad: a BN layer's backward opens with one bn_bwd_reduce_kernel; the igemm / dgrad_s2 launches up to the next one are its data gradient
# (1 launch, or 2 / 4 for the stride-2 layers depending on the tuned form).  The three head convolutions have no BN: the first dgrad launch seen
# while a head layer is next in line belongs to it.
rev = list(reversed(specs))
fwd_ids = {id(r) for r in fwd}
li, got = -1, 0
for r in last:
    nm = r['Kernel_Name']
    if 'bn_bwd_reduce_kernel' in nm:
        li = next(i for i in range(li + 1, len(rev)) if rev[i].bn); got = 0
    elif ('igemm' in nm or 'dgrad_s2_kernel' in nm) and id(r) not in fwd_ids:
        if li + 1 < len(rev) and not rev[li + 1].bn and (li < 0 or got > 0 or not rev[li].bn):
            li += 1; got = 0
        s = rev[li]; got += 1
        agg[(s.cin, s.cout, s.k, s.stride, hw(s))][2] += dur(r)
print(f"{'cin->cout k s @hw':28s} {'n':>3s} {'fwd us':>8s} {'TF':>6s} {'dgrad':>8s} {'TF':>6s} {'wgrad':>8s} {'TF':>6s}  tot ms")
tot = [0, 0, 0]
for k, (n, f, d, w, fl) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2] + kv[1][3])):
    tf = lambda t: fl / t / 1e6 if t else 0
    print(f"{k[0]:4d}->{k[1]:4d} k{k[2]} s{k[3]} @{k[4]:3d}       {n:3d} {f / n:8.1f} {tf(f):6.0f} {d / n:8.1f} {tf(d):6.0f} {w / n:8.1f} {tf(w):6.0f}  {(f + d + w) / 1e3:6.2f}")
    tot[0] += f; tot[1] += d; tot[2] += w
print('totals ms: fwd %.2f dgrad %.2f wgrad %.2f' % tuple(t / 1e3 for t in tot))
stem = collections.OrderedDict()
for r in last:
    if 'stem_' in r['Kernel_Name'] and 'stem_l1_kernel' not in r['Kernel_Name']:
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        stem[n] = stem.get(n, 0.0) + dur(r)
if fused: print('  32->  64 k3 s2 @320 fwd = stem_l1_kernel: stem activation (recomputed from the image) + this convolution + its BN statistics in one launch')
print('stem 3->32 k3 s1 @640 (recompute kernels, us): ' + ', '.join(f'{k} {v:.1f}' for k, v in stem.items()) + f'; total {sum(stem.values()) / 1e3:.2f} ms')
