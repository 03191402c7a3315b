"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs of the same
command, MI355X_MICROARCH.md 'HBM' / 'rocprofv3 PMC slots').  Units: both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B read requests at 64 B, so reads are doubled.  Usage:
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        if name.startswith('void '):
            name = name[5:]
        # template instances stay apart (ncc_cols2_p2<2048> is not ncc_cols2_p2<512>): a kernel's traffic is compared with the
        # algorithmic bytes of the SAME shapes
        name = name.split('(')[0].strip()
        base, targs = (name.split('<', 1) + [''])[:2]
        name = base.split('::')[-1].strip() + ('<' + targs.replace('(anonymous namespace)::', '').replace(' ', '') if targs else '')
        a = acc[name]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, (0, 0.0))
        nw, w = write.get(k, (0, 0.0))
        rd = 2.0 * 1024.0 * f / max(nf, 1)           # gfx950: FETCH_SIZE reads 1/2 of a wide streaming read
        wr = 1024.0 * w / max(nw, 1)
        out[k] = dict(launches=int(max(nf, nw)), read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                      hbm_bytes_per_launch=rd + wr)
    json.dump(dict(note='FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE, KiB -> bytes, mean per launch; separate --pmc passes; keyed by template instance',
                   kernels=out), open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:14]:
        print(f"{k[:44]:44s} n={v['launches']:5d} rd={v['read_bytes_per_launch']/1e6:10.2f} MB wr={v['write_bytes_per_launch']/1e6:10.2f} MB")


if __name__ == '__main__':
    main()
