"""tools/pmc_bwd.sh summary (rocprofv3 --pmc, four counter sets in separate runs, kernel trace only) -> per-kernel derived
figures: MFMA utilisation, wave-time split, L1 / L2 hit rates, LDS bank-conflict share.
    python tools/pmc_summary.py gpurun_out/pmc_bwd/summary.txt profiles/r06_pmc_gemm   (writes .txt and .json)
Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_16x16x4_f32),
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_BUSY_CYCLES is summed over the 32
shader engines, TCP_* / TCC_* are summed over the chip."""
import collections
import json
import sys

N_SIMD, N_SE, CLOCK_HZ = 1024, 32, 2.4e9


def parse(path):
    out = collections.defaultdict(dict)
    kern = None
    for line in open(path):
        line = line.rstrip()
        if line.startswith('=='):
            continue
        if line.startswith('    '):
            name, _, val = line.split()
            out[kern][name] = float(val)
        elif line.strip():
            kern = line.split('(')[0].strip()
            out[kern]['dispatches'] = float(line.split()[-1])
    return out


def derive(c):
    d = {}
    busy_per_se = c['SQ_BUSY_CYCLES'] / N_SE                        # cycles some wave was resident, per shader engine
    d['kernel_busy_us'] = round(busy_per_se / CLOCK_HZ * 1e6, 2)
    d['mfma_instructions'] = int(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 32)
    d['mfma_gflop'] = round(d['mfma_instructions'] * 2048 / 1e9, 4)
    d['mfma_util_of_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (N_SIMD * busy_per_se), 4)
    d['wave_time_waiting_for_any_instruction'] = round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4)
    d['wave_time_waiting_any'] = round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4)
    d['wave_time_issuing'] = round(c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4)
    d['waves'] = int(c['SQ_WAVES'])
    d['l2_hit_rate'] = round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 4)
    d['l1_hit_rate'] = round(1 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum'], 4)
    d['l1_to_l2_read_requests'] = int(c['TCP_TCC_READ_REQ_sum'])
    d['l1_to_l2_read_MB_at_64B'] = round(c['TCP_TCC_READ_REQ_sum'] * 64 / 1e6, 1)
    d['lds_bank_conflict_share'] = round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4)
    d['vmem_read_instructions'] = int(c['SQ_INSTS_VMEM_RD'])
    d['valu_instructions'] = int(c['SQ_INSTS_VALU'])
    d['lds_instructions'] = int(c['SQ_INSTS_LDS'])
    return d


if __name__ == '__main__':
    src, dst = sys.argv[1], sys.argv[2]
    raw = parse(src)
    res = {k: {'derived': derive(v), 'counters': v} for k, v in raw.items()}
    with open(dst + '.json', 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
    with open(dst + '.txt', 'w') as f:
        f.write(__doc__.split('Units')[0].strip() + '\n')
        f.write(f'# source: {src} (MM-IMDB b128, eager steps of bench.py; production launch, BMNAS_CONV_PROBE = 0)\n')
        for k, v in res.items():
            f.write(f'\n{k}\n')
            for n, x in v['derived'].items():
                f.write(f'    {n:42s} {x}\n')
            f.write('    -- raw averages per dispatch\n')
            for n, x in sorted(v['counters'].items()):
                f.write(f'    {n:42s} {x:.1f}\n')
    print(open(dst + '.txt').read())
