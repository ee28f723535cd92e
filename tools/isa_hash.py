#!/usr/bin/env python3
"""Per-kernel hash of the gfx950 machine code inside a built library / object (instruction text without addresses), to show that a
refactor left a kernel's code untouched:   tools/isa_hash.py lib.so [name-filter ...]      (compare two outputs with diff)"""
import hashlib
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def code_objects(path):
    """the gfx950 code objects bundled in a host library / object (one fat binary per translation unit, concatenated in
    section .hip_fatbin)"""
    fat = tempfile.mktemp(suffix='.fat')
    subprocess.check_call([LLVM + '/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat])
    blob = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    outs = []
    for i, s0 in enumerate(starts):
        part = tempfile.mktemp(suffix='.bundle')
        open(part, 'wb').write(blob[s0:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
        out = tempfile.mktemp(suffix='.co')
        subprocess.check_call([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + part,
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + out])
        outs.append(out)
    return outs


def main():
    filters = sys.argv[2:]
    dis = ''.join(subprocess.run([LLVM + '/llvm-objdump', '-d', '--no-show-raw-insn', co], capture_output=True, text=True).stdout
                  for co in code_objects(sys.argv[1]))
    cur, body, rows = None, [], []

    def flush():
        if cur and body:
            name = subprocess.run(['c++filt', cur], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'^void scs::|scs::|\(.*', '', name)
            if not filters or all(f in name for f in filters):
                rows.append((name, len(body), hashlib.sha1('\n'.join(body).encode()).hexdigest()[:12]))

    for ln in dis.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
        if m:
            flush()
            cur, body = m.group(1), []
        elif cur and ln[:1] in (' ', '\t'):
            t = re.sub(r'//.*', '', ln).strip()
            t = re.sub(r'^[0-9a-f]+:\s*', '', t)
            if t:
                body.append(t)
    flush()
    for r in sorted(rows):
        print('%-140s %6d %s' % (r[0][:140], r[1], r[2]))


if __name__ == '__main__':
    main()
