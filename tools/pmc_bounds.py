"""Merge the per-kernel counter passes of tools/collect_counters.sh (gpurun_out/<tag>_pmc_{lds1,lds2,sq3,tcc1,tcc2,tcc3}.json) into
ONE table of what bounds each MFMA kernel, with the derived fractions next to the raw means per dispatch:

    python tools/pmc_bounds.py r05 > profiles/r05_pmc_bounds.json

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "Per-instruction cycle constants"): GRBM_GUI_ACTIVE is summed over the 8 XCDs
(cycles of one XCD = / 8); SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES
and SQ_LDS_IDX_ACTIVE count cycles summed over SIMDs / CUs; TCC_EA0_RDREQ_{32,64,128}B are the L2 -> fabric read requests by size.
"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
SIMDS, CUS, XCDS = 1024, 256, 8
allk = {}
for p in ('lds1', 'lds2', 'sq3', 'tcc1', 'tcc2', 'tcc3'):
    try:
        d = json.load(open(f'gpurun_out/{tag}_pmc_{p}.json'))
    except OSError:
        continue
    for k, v in d.items():
        allk.setdefault(k, {}).update({c: x['mean'] for c, x in v.items()})
out = {'note': __doc__.strip().split('\n\n')[-1].replace('\n', ' ')}
for k, c in allk.items():
    cyc = c['GRBM_GUI_ACTIVE'] / XCDS                      # shader cycles of the launch
    n_mfma = c.get('SQ_INSTS_MFMA', 0)
    instr = c.get('SQ_INSTS_VALU', 0) + c.get('SQ_INSTS_LDS', 0) + c.get('SQ_INSTS_SALU', 0) + c.get('SQ_INSTS_VMEM_RD', 0)
    d = {
        'cycles_per_launch': round(cyc),
        'mfma_busy': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (SIMDS * cyc), 4),
        'lds_array_busy': round(c['SQ_LDS_IDX_ACTIVE'] / (CUS * cyc), 4),
        'lds_bank_conflict_frac_of_lds_cycles': round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), 4),
        'valu_busy_incl_mfma_issue': round(4 * c['SQ_ACTIVE_INST_VALU'] / (SIMDS * cyc), 4),
        'wave_time_parked': round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4),
        'wave_time_issue_stalled': round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4),
        'wave_time_issue_stalled_on_lds': round(c['SQ_WAIT_INST_LDS'] / c['SQ_WAVE_CYCLES'], 4),
        'wave_time_issuing': round(c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4),
        'instructions_per_mfma': round(instr / max(n_mfma, 1), 2),
        'valu_per_mfma': round((c.get('SQ_INSTS_VALU', 0) - n_mfma) / max(n_mfma, 1), 2),
        'lds_per_mfma': round(c.get('SQ_INSTS_LDS', 0) / max(n_mfma, 1), 2),
        'salu_per_mfma': round(c.get('SQ_INSTS_SALU', 0) / max(n_mfma, 1), 2),
        'l2_hit_rate': round(c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1), 4),
        'fabric_read_bytes': int(32 * c['TCC_EA0_RDREQ_32B_sum'] + 64 * c['TCC_EA0_RDREQ_64B_sum'] + 128 * c['TCC_EA0_RDREQ_128B_sum']),
        'fabric_write_bytes_64B_requests': int(64 * c['TCC_EA0_WRREQ_64B_sum']),
        'l2_read_requests': int(c['TCC_READ_sum']),
        'raw': {n: round(x, 1) for n, x in sorted(c.items())},
    }
    out[k] = d
print(json.dumps(out, indent=1))
