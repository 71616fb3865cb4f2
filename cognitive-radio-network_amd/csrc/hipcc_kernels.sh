#!/bin/bash
# hipcc_kernels.sh <out.o> <src.hip> <hipcc flags...> — what `hipcc -c` does for one HIP translation unit, with one step in between:
# the gfx950 assembly passes through strip_asm_nops.py (the wait states hipcc puts between this library's inline-asm packed-f32
# instructions: see that file).  Steps = hipcc's own (`hipcc -###`): device code to assembly, assemble, link the code object,
# bundle it (compressed), compile the host side around the bundle.  $CRN_KEEP_ASM_NOPS=1 builds the plain way instead.
set -euo pipefail
out=$1; src=$2; shift 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if [ "${CRN_KEEP_ASM_NOPS:-0}" = "1" ]; then exec "$HIPCC" "$@" -c -o "$out" "$src"; fi
LLVM=${ROCM_LLVM_BIN:-$(dirname "$(readlink -f "$HIPCC")")/../lib/llvm/bin}
ARCH=${ARCH:-gfx950}
here=$(dirname "$(readlink -f "$0")")
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
"$HIPCC" "$@" -Wno-unused-command-line-argument --cuda-device-only -S -o "$tmp/dev.s" "$src"
python3 "$here/strip_asm_nops.py" "$tmp/dev.s" "$tmp/dev_f.s" | sed "s|^|$(basename "$src"): |"
"$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu="$ARCH" -c "$tmp/dev_f.s" -o "$tmp/dev.o"
"$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$tmp/dev.out" "$tmp/dev.o"
"$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--"$ARCH" \
  -input=/dev/null -input="$tmp/dev.out" -output="$tmp/dev.hipfb" --compress
"$HIPCC" "$@" -Wno-unused-command-line-argument --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$tmp/dev.hipfb" -c -o "$out" "$src"
