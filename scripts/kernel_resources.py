"""Per-kernel register / LDS / scratch figures of the built HIP objects (catgrasp_amd/csrc/*.o), read from the code-object metadata:
.hip_fatbin section -> clang-offload-bundler (gfx950 image) -> llvm-readelf --notes.  Prints CSV; `resources()` is what the CPU test uses."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'catgrasp_amd', 'csrc')
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count',
          'max_flat_workgroup_size')


def demangle(names):
    import shutil
    tool = shutil.which('c++filt')
    if tool is None:
        return list(names)
    out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
    return [o.replace('(anonymous namespace)::', '') for o in out[:len(names)]]


def resources(obj):
    """-> [{'kernel': demangled name, vgpr_count: ..., ...}] of one host object with an embedded gfx950 image."""
    with tempfile.TemporaryDirectory() as d:
        fat, dev = os.path.join(d, 'fat'), os.path.join(d, 'dev.o')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), f'--dump-section=.hip_fatbin={fat}', obj], check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={fat}',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={dev}'], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', dev], check=True, capture_output=True, text=True).stdout
    rows, cur = [], None
    for line in notes.split('\n'):
        if re.match(r'  - \.\w+:', line):                       # a kernel's block of the `amdhsa.kernels` list (keys sorted: .agpr_count first)
            cur = {}
            rows.append(cur)
            line = '    ' + line[4:]
        m = re.match(r'    \.(\w+):\s*(.*)$', line)
        if not m or cur is None:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == 'name':
            cur['kernel'] = v
        elif k in FIELDS:
            cur[k] = int(v)
    rows = [r for r in rows if 'vgpr_count' in r and 'kernel' in r]
    for r, n in zip(rows, demangle([r['kernel'] for r in rows])):
        r['kernel'] = re.sub(r'\(.*$', '', n)
    return rows


if __name__ == '__main__':
    print('object,kernel,' + ','.join(FIELDS))
    for obj in sorted(glob.glob(os.path.join(CSRC, '*.o'))):
        try:
            for r in resources(obj):
                print(','.join([os.path.basename(obj), '"' + r['kernel'] + '"'] + [str(r.get(f, '')) for f in FIELDS]))
        except subprocess.CalledProcessError:
            print(f'{os.path.basename(obj)},(no gfx950 image),' + ',' * (len(FIELDS) - 1), file=sys.stderr)
