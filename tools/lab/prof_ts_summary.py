"""Median duration per (kernel, grid) of the TN kernels in a rocprofv3 --kernel-trace csv directory."""
import sys, csv, glob, collections
d, tag = sys.argv[1], sys.argv[2]
fs = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
r = collections.defaultdict(list)
for f in fs:
    for row in csv.DictReader(open(f)):
        n = row['Kernel_Name']
        if 'split_tn' in n or 'gemm_ring' in n or 'splitk' in n:
            r[(n[:40], row.get('Grid_Size_X', row.get('Grid_Size', '?')))].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for k, v in sorted(r.items()):
    v.sort()
    print(f'{tag:10s} {k[0]:40s} grid {k[1]:>8s} n {len(v):3d} median {v[len(v) // 2] / 1000:7.1f} us  min {v[0] / 1000:7.1f}')
