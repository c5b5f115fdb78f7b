"""Turn per-kernel rocprofv3 PMC summaries (tools/pmc_summary.py output of a FETCH_SIZE pass and of a WRITE_SIZE pass over
bench.py) into profiles/pmc_current.json: corrected HBM bytes per launch of the two roofline kernels, stamped with the sha256 of
the kernel sources and the git commit they were measured at.  bench.py quotes `roofline.traffic` from that file and nulls it
when the stamp no longer matches the tree.  Run in the build container after the GPU call:

    python tools/pmc_roofline.py gpurun_out/rNN_pmc_fetch.json gpurun_out/rNN_pmc_write.json [gpurun_out/rNN_pmc_sq.json]

The optional third file is an SQ pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, ...) of the same command: it adds `mfma_busy` =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) per kernel, which bench.py quotes beside `frac`.  An optional
fourth file is the LDS / VALU pass of tools/collect_counters.sh (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_ACTIVE_INST_VALU,
SQ_ACTIVE_INST_LDS, ...): it adds `lds_busy` = SQ_LDS_IDX_ACTIVE (LDS-array cycles, MI355X_MICROARCH.md LDS section) / (256 CUs x
GRBM_GUI_ACTIVE / 8) and `valu_busy` = 4 x SQ_ACTIVE_INST_VALU (quad-cycles, MFMA issue included) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests as 64 bytes -> read bytes =
2 x FETCH_SIZE(KiB) x 1024 for the wide coalesced rows these kernels stream; WRITE_SIZE(KiB) x 1024 as is.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import PMC_FILE, source_stamp  # noqa: E402

KERNELS = {'window_attn_kernel': ['window_attn.hip', 'common.h'], 'gsv3_kernel': ['global_match.hip', 'common.h'],
           'gsv4_kernel': ['global_match.hip', 'common.h'], 'ffn_kernel': ['ffn.hip', 'common.h'], 'kv4_kernel': ['ffn.hip', 'common.h']}
NUM_SIMDS, NUM_XCDS, NUM_CUS = 1024, 8, 256


def main():
    git = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    dirty = bool(subprocess.run(['git', 'status', '--porcelain', 'unimatch_amd/csrc'], capture_output=True, text=True,
                                cwd=ROOT).stdout.strip())
    out = {'git': git + ('+dirty' if dirty else ''),
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) over `python bench.py '
                   '--steps 3 --warmup 2 --no-cpu-baseline`; read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 correction), write '
                   'bytes = WRITE_SIZE KiB x 1024; means per dispatch'}
    if '--from-r01' in sys.argv:
        old = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_final.json')))
        for k, v in old.items():
            base = k.split('<')[0]
            if base in KERNELS:
                out[k] = {'hbm_traffic_bytes_per_launch': v['hbm_traffic_bytes_per_launch'], 'FETCH_SIZE_KiB': v['FETCH_SIZE_KiB'],
                          'WRITE_SIZE_KiB': v['WRITE_SIZE_KiB'], 'source_stamp': source_stamp(KERNELS[base]),
                          'from': 'profiles/r01_pmc_final.json'}
    else:
        files = [a for a in sys.argv[1:] if not a.startswith('--')]
        fetch, write = json.load(open(files[0])), json.load(open(files[1]))
        sq = json.load(open(files[2])) if len(files) > 2 else {}
        lds = json.load(open(files[3])) if len(files) > 3 else {}
        for k, v in fetch.items():
            base = k.split('<')[0]
            if base not in KERNELS or 'FETCH_SIZE' not in v:
                continue
            f = v['FETCH_SIZE']['mean']
            w = write.get(k, {}).get('WRITE_SIZE', {}).get('mean', 0.0)
            out[k] = {'hbm_traffic_bytes_per_launch': int(2 * f * 1024 + w * 1024), 'FETCH_SIZE_KiB': round(f, 1),
                      'WRITE_SIZE_KiB': round(w, 1), 'dispatches_sampled': v['FETCH_SIZE']['dispatches'],
                      'source_stamp': source_stamp(KERNELS[base])}
            c = sq.get(k, {})
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
                busy, gui = c['SQ_VALU_MFMA_BUSY_CYCLES']['mean'], c['GRBM_GUI_ACTIVE']['mean']
                out[k].update({'mfma_busy': round(busy / (NUM_SIMDS * gui / NUM_XCDS), 4), 'SQ_VALU_MFMA_BUSY_CYCLES': busy,
                               'GRBM_GUI_ACTIVE': gui,
                               **{n: v['mean'] for n, v in c.items() if n not in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE')}})
            c = lds.get(k, {})
            if 'SQ_LDS_IDX_ACTIVE' in c and 'GRBM_GUI_ACTIVE' in c:
                gui = c['GRBM_GUI_ACTIVE']['mean'] / NUM_XCDS
                out[k].update({'lds_busy': round(c['SQ_LDS_IDX_ACTIVE']['mean'] / (NUM_CUS * gui), 4),
                               'valu_busy': round(4 * c['SQ_ACTIVE_INST_VALU']['mean'] / (NUM_SIMDS * gui), 4),
                               'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT', {}).get('mean'),
                               **{n: v['mean'] for n, v in c.items() if n in ('SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
                                                                              'SQ_INSTS_LDS')}})
    json.dump(out, open(PMC_FILE, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
