"""idle gaps of the GPU timeline from a rocprofv3 --kernel-trace csv: which kernels are followed by idle time"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]) for r in rows))
# restrict to the last 60 % of the run (steady state)
t_lo = ev[0][0] + 0.4 * (ev[-1][1] - ev[0][0])
ev = [e for e in ev if e[0] >= t_lo]
busy = 0; gaps = collections.Counter(); gapn = collections.Counter(); end = ev[0][0]; tot_gap = 0
for s, e, k in ev:
    if s > end:
        g = s - end; tot_gap += g; gaps[prev] += g; gapn[prev] += 1
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end: end = e; prev = k
span = ev[-1][1] - ev[0][0]
print(f'span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %) idle {tot_gap/1e6:.2f} ms')
for k, g in gaps.most_common(12):
    print(f'  idle after {k[:40]:40s} {g/1e6:8.3f} ms in {gapn[k]:5d} gaps ({g/gapn[k]/1e3:7.1f} us each)')
