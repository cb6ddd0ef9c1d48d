"""Summarise a rocprofv3 --pmc pass of SQ counters (csv output: counter_collection.csv with Kernel_Name, Counter_Name,
Counter_Value per dispatch) into per-kernel fractions of SQ_WAVE_CYCLES, as profiles/r02_pmc_sq_wave_cycles.json does.
usage: pmc_sq_summary.py counter_collection.csv out.json [kernel-name filter]"""
import collections
import csv
import json
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            n[k] += 1
    flt = sys.argv[3] if len(sys.argv) > 3 else ''
    out = {}
    for k, c in sorted(acc.items()):
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        if not wc or flt not in k:
            continue
        out[k] = {'dispatches': n[k], 'wave_cycles_per_dispatch': wc / max(n[k], 1),
                  'waiting (s_waitcnt / barrier)': round(c.get('SQ_WAIT_ANY', 0.0) / wc, 3),
                  'issue stall': round(c.get('SQ_WAIT_INST_ANY', 0.0) / wc, 3),
                  'issuing any': round(c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 3),
                  'issuing VALU': round(c.get('SQ_ACTIVE_INST_VALU', 0.0) / wc, 3),
                  'issuing LDS': round(c.get('SQ_ACTIVE_INST_LDS', 0.0) / wc, 3),
                  'lds_bank_conflict_per_lds_cycle': round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0), 4)}
        print(k, json.dumps(out[k]))
    json.dump({'source': 'rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU '
                         'SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE', 'kernels': out}, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
