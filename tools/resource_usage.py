#!/usr/bin/env python
"""Register / scratch / occupancy table of the kernels of one translation unit, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles for gfx950 without a GPU).
usage: python tools/resource_usage.py qcqp_amd/csrc/cd_queue.hip [substring of the kernel name ...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(unit, extra=()):
    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
           '-I' + os.path.join(REPO, 'include'), '-Rpass-analysis=kernel-resource-usage', '-c', unit, '-o', '/dev/null'] + list(extra)
    err = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE).stderr.decode()
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark:\s+(.*?)\s+\[-Rpass-analysis', line)
        if not m:
            continue
        text = m.group(1)
        if text.startswith('Function Name:'):
            name = text.split(':', 1)[1].strip()
            try:
                name = subprocess.run(['c++filt', name], stdout=subprocess.PIPE).stdout.decode().strip() or name
            except OSError:
                pass
            cur = {'name': name}
            rows.append(cur)
        elif cur is not None and ':' in text:
            k, v = text.split(':', 1)
            cur[k.strip()] = v.strip()
    return rows


def main():
    unit = sys.argv[1]
    pats = sys.argv[2:]
    keys = ['VGPRs', 'AGPRs', 'TotalSGPRs', 'VGPRs Spill', 'SGPRs Spill', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]']
    print('| kernel | ' + ' | '.join(keys) + ' |')
    print('|---|' + '---|' * len(keys))
    for r in table(unit):
        short = r['name'].replace('qcqpmi::(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\(qcqpmi::.*$', '', short)
        if pats and not any(p in short for p in pats):
            continue
        print('| `%s` | ' % short + ' | '.join(r.get(k, '?') for k in keys) + ' |')


if __name__ == '__main__':
    main()
