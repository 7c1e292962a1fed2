"""What hipcc made of a kernel: loads, waits, barriers, branches and MFMAs in program order.

    python tools/isa_trace.py conv1x1 conv_bwd_all_pipe_kILi48ELi3ELi2      # csrc/<name>.hip, mangled-name substring
    python tools/isa_trace.py bnmix node_mix_ln_bwd_k --loops                # one line per loop instead

Round 2 found most of its late gains here ("What hipcc made of the latency-sensitive
bodies"): a kernel-argument pointer array indexed at run time (a memory load of the pointer + s_waitcnt vmcnt(0)
per block), loads under `if` (register copies behind a wait at the join), a fold written at the point of load
(the software pipeline existed in the source only), a skippable stash (its loads pending on one path = a drain at
the loop head on every path).  None of them shows in the source; all of them show as `L.. w0` patterns here.

Legend: L global/flat load, S store, A atomic, r / W LDS read / write, M MFMA, |B| s_barrier, br branch,
wN s_waitcnt vmcnt(N), X scratch access.  Runs of the same token are compressed (L12 = twelve loads).
No GPU needed (hipcc cross-compiles gfx950)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-pass-failed', '-S',
         '--cuda-device-only']


def token(x):
    if x.startswith(('global_load', 'flat_load', 'buffer_load')):
        return 'L'
    if x.startswith(('global_store', 'flat_store', 'buffer_store')):
        return 'S'
    if x.startswith(('global_atomic', 'flat_atomic', 'buffer_atomic')):
        return 'A'
    if x.startswith('scratch_'):
        return 'X'
    if x.startswith('ds_read'):
        return 'r'
    if x.startswith('ds_write'):
        return 'W'
    if x.startswith('v_mfma'):
        return 'M'
    if x.startswith('s_barrier'):
        return '|B|'
    if x.startswith('s_cbranch'):
        return 'br'
    if x.startswith('s_waitcnt'):
        m = re.search(r'vmcnt\((\d+)\)', x)
        if m:
            return 'w' + m.group(1)
    return None


def compress(tokens):
    out = []
    for t in tokens:
        if out and out[-1][0] == t:
            out[-1][1] += 1
        else:
            out.append([t, 1])
    return ' '.join(f'{t}{n if n > 1 else ""}' for t, n in out)


def kernels(asm):
    name, body = None, []
    for line in asm.split('\n'):
        m = re.match(r'^(_Z\S+):', line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if 's_endpgm' in line:
                yield name, body
                name, body = None, []


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        return 2
    src = os.path.join(ROOT, 'bm-nas_amd', 'csrc', sys.argv[1] + '.hip')
    pat = sys.argv[2]
    loops = '--loops' in sys.argv[3:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [src, '-o', out], check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    meta = {}
    cur = None
    for line in asm.split('\n'):
        m = re.match(r'^\s+\.amdhsa_kernel (\S+)', line)
        if m:
            cur = m.group(1)
            meta[cur] = {}
        for key in ('next_free_vgpr', 'private_segment_fixed_size', 'group_segment_fixed_size'):
            m = re.match(r'^\s+\.amdhsa_%s (\d+)' % key, line)
            if m and cur:
                meta[cur][key] = int(m.group(1))
    found = 0
    for name, body in kernels(asm):
        if pat not in name:
            continue
        found += 1
        print(f'== {name[:110]}')
        print(f'   {len(body)} lines, {meta.get(name, {})}')
        if not loops:
            print('   ' + compress([t for t in (token(x.strip()) for x in body) if t]))
            continue
        labels = {}
        for i, line in enumerate(body):
            m = re.match(r'^(\.LBB\S+):', line)
            if m:
                labels[m.group(1)] = i
        for i, line in enumerate(body):
            m = re.search(r's_cbranch\S+\s+(\.LBB\S+)', line)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = body[labels[m.group(1)]:i + 1]
                toks = [t for t in (token(x.strip()) for x in seg) if t]
                if any(t in ('L', 'M') for t in toks):
                    print(f'   loop {m.group(1)} ({len(seg)} lines): ' + compress(toks))
    if not found:
        print(f'no kernel matching {pat!r} in {src}')
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
