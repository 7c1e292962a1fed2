"""FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section: widths other than 16 B/lane are uncalibrated).

  python tools/calibrate_fetch.py run            # launches the probe reads (run it under rocprofv3 --pmc)
  python tools/calibrate_fetch.py report <counter_collection.csv> <out.txt>

`run` reads a fresh 256 MiB buffer once per launch with 4 / 8 / 16 bytes per lane and as 64-byte rows at
row strides of 16 floats (dense: the split-K kernels' operand rows) and 32 floats (every other 64 B).
`report` divides the known byte counts by the counter (KiB units)."""
import csv
import sys

N = 64 * 1024 * 1024            # floats = 256 MiB
CASES = [(4, 0), (8, 0), (16, 0), (64, 16), (64, 32)]
KERNEL = {4: 'probe_read_k<float>', 8: 'probe_read_k<float2>', 16: 'probe_read_k<float4>', 64: 'probe_rows_k'}


def run():
    import torch
    sys.path.insert(0, 'bm-nas_amd')
    from bmnas import lib
    sink = torch.zeros(4, device='cuda')
    for rep in range(3):
        for width, ld in CASES:
            buf = torch.full((N,), 1.0, device='cuda')       # freshly written: not resident in any L2 as clean lines
            flush = torch.full((N,), 2.0, device='cuda')     # push it out of the 256 MiB Infinity Cache as well
            del flush
            torch.cuda.synchronize()
            lib.probe_read(buf, width, sink, ld)
            torch.cuda.synchronize()
            del buf


def report(path, out):
    rows = {}
    order = []
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != 'FETCH_SIZE' or 'probe_' not in r['Kernel_Name']:
            continue
        order.append((r['Kernel_Name'], float(r['Counter_Value'])))
    # launches come in CASES order, three repetitions
    lines = ['# FETCH_SIZE calibration on gfx950: bytes actually read / (FETCH_SIZE * 1024); 256 MiB buffer, read once',
             '# pattern                                 bytes_read     FETCH_SIZE(KiB)   correction']
    for i, (width, ld) in enumerate(CASES):
        vals = [v for j, (_, v) in enumerate(order) if j % len(CASES) == i]
        if not vals:
            continue
        read = N * 4 if ld in (0, 16) else N * 4 // (ld // 16)
        mean = sum(vals) / len(vals)
        name = f'{width} B/lane coalesced' if ld == 0 else f'64-B rows, row stride {ld * 4} B (dword/lane)'
        lines.append(f'{name:42s} {read:12d} {mean:16.0f}   x{read / (mean * 1024):.3f}')
        rows[(width, ld)] = read / (mean * 1024)
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        report(sys.argv[2], sys.argv[3])
