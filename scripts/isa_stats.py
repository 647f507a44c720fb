#!/usr/bin/env python3
"""Instruction statistics of one kernel in a hipcc -S listing: opcode class counts, spill / barrier / atomic positions.

usage: isa_stats.py file.s <substring of the mangled kernel name> [--dump]
"""
import collections
import re
import sys


def kernel_body(text, key):
    lines = text.split('\n')
    start = None
    for i, l in enumerate(lines):
        if start is None and l.startswith('_Z') and key in l.split(':')[0] and ':' in l:
            start = i
        if start is not None and 's_endpgm' in l and i > start:
            return lines[start:i + 1]
    return None


def main():
    text = open(sys.argv[1]).read()
    body = kernel_body(text, sys.argv[2])
    if body is None:
        sys.exit('kernel not found')
    cnt = collections.Counter()
    for l in body:
        t = l.strip().split()
        if not t or t[0].endswith(':') or t[0].startswith('.') or t[0].startswith(';'):
            continue
        op = t[0]
        cls = ('scratch' if op.startswith('scratch_') else 'ds_add' if op.startswith('ds_add') else 'ds_other' if op.startswith('ds_') else
               'global' if op.startswith('global_') or op.startswith('buffer_') else 'valu' if op.startswith('v_') else
               'salu' if op.startswith('s_') else 'other')
        cnt[cls] += 1
        if op in ('s_barrier', 's_waitcnt', 'v_readlane_b32', 'v_writelane_b32'):
            cnt[op] += 1
    print(len(body), 'lines', dict(cnt))
    print('scratch at', [i for i, l in enumerate(body) if 'scratch_' in l])
    adds = [i for i, l in enumerate(body) if 'ds_add_f64' in l]
    print('ds_add_f64:', len(adds), 'first', adds[:3], 'last', adds[-3:])
    print('barriers at', [i for i, l in enumerate(body) if 's_barrier' in l])
    if '--dump' in sys.argv:
        for i, l in enumerate(body):
            print(i, l)


if __name__ == '__main__':
    main()
