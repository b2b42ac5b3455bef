#!/usr/bin/env python3
"""MFMA utilisation and wave-stall breakdown of the gate GEMM + cell launches from one rocprofv3 PMC pass (SQ counters).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \\
              SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/pmc_g -o r \\
              -- python3 bench.py --steps 2 --warmup 3 --frozen-steps 0 --no-cpu-baseline --no-roofline
    python profiles/pmc_gemm.py gpurun_out/pmc_g > profiles/r02_pmc_gemm.json

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (= 64 x the number
of v_mfma_f32_32x32x2_f32 wave instructions); SQ_BUSY_CYCLES is summed over the 32 shader engines, so kernel cycles =
SQ_BUSY_CYCLES / 32; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave and are used as ratios only.
"""
import collections
import csv
import glob
import json
import sys

N_SIMD, N_SE = 1024, 32


def kernel_key(full):
    """'void (anonymous namespace)::k_gate_cell_p<4, 4, 8>((anonymous namespace)::GemmArgs, int)' -> 'k_gate_cell_p<4, 4, 8>'
    (round 2 split at '::' and kept the tail of the ARGUMENT list, 'GemmArgs, int)')."""
    import re
    m = re.search(r'(k_\w+(?:<[^>]*>)?)', full)
    return m.group(1) if m else full.split('(')[0]


def main(d, pat):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            name = kernel_key(r['Kernel_Name'])
            per[name][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {'source': 'rocprofv3 --pmc (SQ counters, one pass, --kernel-trace only) -- python3 bench.py --steps 2 --warmup 3 '
                     '--frozen-steps 0 --no-cpu-baseline --no-roofline', 'kernels': {}}
    tot_mfma = tot_cyc = tot_wave = tot_wait = tot_stall = tot_act = 0.0
    for name, cs in sorted(per.items()):
        avg = {k: sum(v) / len(v) for k, v in cs.items()}
        cyc = avg['SQ_BUSY_CYCLES'] / N_SE
        rec = {'dispatches': len(cs['SQ_BUSY_CYCLES']), 'kernel_cycles': round(cyc),
               'mfma_busy_cycles_per_simd': round(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / N_SIMD),
               'mfma_busy_frac': round(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / N_SIMD / cyc, 3),
               'wave_wait_memory_frac': round(avg['SQ_WAIT_ANY'] / avg['SQ_WAVE_CYCLES'], 3),
               'wave_issue_stall_frac': round(avg['SQ_WAIT_INST_ANY'] / avg['SQ_WAVE_CYCLES'], 3),
               'wave_active_frac': round(avg['SQ_ACTIVE_INST_ANY'] / avg['SQ_WAVE_CYCLES'], 3),
               'mfma_mops_f32': round(avg.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0))}
        out['kernels'][name] = rec
        n = rec['dispatches']
        tot_mfma += avg['SQ_VALU_MFMA_BUSY_CYCLES'] * n
        tot_cyc += cyc * n
        tot_wave += avg['SQ_WAVE_CYCLES'] * n
        tot_wait += avg['SQ_WAIT_ANY'] * n
        tot_stall += avg['SQ_WAIT_INST_ANY'] * n
        tot_act += avg['SQ_ACTIVE_INST_ANY'] * n
    out['mfma_busy_frac'] = round(tot_mfma / N_SIMD / tot_cyc, 3)
    out['wave_stall_frac'] = {'waiting_on_memory_or_barrier': round(tot_wait / tot_wave, 3),
                              'issue_stalled_mfma_dependency_or_pipe': round(tot_stall / tot_wave, 3),
                              'issuing': round(tot_act / tot_wave, 3)}
    busy = out.get('mfma_busy_frac', 0.0)
    if busy > 0:
        out['reading'] = (f'the fp32 MFMA pipe is busy for {busy:.2f} of the launch and no wave class is saturated: the launch is the sum of '
                          'its phases (operand latency, MFMA chain, cell arithmetic on the same vector pipe, store drain) rather than their '
                          'maximum -- see HISTORY.md section C')
    elif pat.startswith('k_spmm'):
        out['reading'] = ('no MFMA work and no LDS: a thread walks the dependent chain row pointer / ELL entry -> neighbour rows (gathers through '
                          'L1 / L2) -> one output row; waves wait on those global loads for most of the launch (waiting_on_memory_or_barrier) '
                          'and issue vector instructions for the rest -- the launch is bound by the memory system\'s latency x requests in '
                          'flight, cf. the PMC traffic record of the same run')
    elif pat.startswith('k_remesh'):
        out['reading'] = ('no MFMA work, 2 x 1024 threads per CU: stage a 4-channel source slice in LDS (global loads), gather it per pixel '
                          '(LDS), lane-group sums, scattered 16-byte row stores.  Waves are issue-stalled -- instructions ready but the LDS / '
                          'vector-memory queues of the CU full -- for about half of the launch and wait on memory or one of the three '
                          'barriers for most of the rest; they issue for < 10 %.  The launch is one workgroup\'s chain of dependent phases, '
                          'not a bandwidth stream: PMC traffic = its operands once')
    elif pat.startswith('k_attn'):
        out['reading'] = ('no MFMA work: one node per lane group, online softmax over gathered k / v rows; waves wait on the gathers of the '
                          'neighbour rows (global memory) for most of the launch')
    else:
        out['reading'] = ('no MFMA work: waves wait on LDS / barriers / the prologue\'s loads for more than half of the launch and issue '
                          'vector instructions for the rest (a wave64 instruction occupies a SIMD for 4 cycles) -- see HISTORY.md section C')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'k_gate_cell_p')
