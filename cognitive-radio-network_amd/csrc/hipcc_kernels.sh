#!/bin/bash
# hipcc_kernels.sh <out.o> <src.hip> <hipcc flags...> — `hipcc -c` for one HIP translation unit with one step in between: the gfx950
# assembly passes through strip_asm_nops.py (the wait states hipcc puts between this library's inline-asm packed-f32 instructions: see
# that file).  Steps = hipcc's own (`hipcc -###`): device code to assembly, assemble, link the code object, bundle it (compressed),
# compile the host side around the bundle.
#
# The filter is only trusted where it was checked.  The unit is built by plain `hipcc -c` instead — with a warning on stderr, never a
# failed build — when
#   * $CRN_KEEP_ASM_NOPS=1 asks for it,
#   * the target is not gfx950 or the compiler is not one of asm_nops.expected's `toolchain` lines (the argument for dropping the wait
#     states is about this LLVM's hazard recogniser on this architecture),
#   * any hand-made step fails (another ROCm lays its tools out differently),
#   * the filter finds no inline-asm markers at all, or removes / keeps a different number of wait states than asm_nops.expected records
#     for this unit (the kernels or the compiler's output changed: look, then `make -C csrc expected`).
# <out.o>.how records which way the unit was built and the counts.  $CRN_KEEP_ASM=<dir> keeps the assembly before and after the filter.
set -uo pipefail
out=$1; src=$2; shift 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${ARCH:-gfx950}
here=$(dirname "$(readlink -f "$0")")
key=${CRN_ASM_EXPECT_KEY:-$(basename "$src")}
expected=${CRN_ASM_EXPECTED:-$here/asm_nops.expected}   # (the override is for tests/test_asm_filter.py)

build_plain() {
  local why=$1; shift
  if [ "$why" != "asked for (CRN_KEEP_ASM_NOPS=1)" ]; then
    echo "hipcc_kernels.sh: WARNING: $(basename "$src") is built by plain hipcc -c, WITHOUT the assembly filter: $why" >&2
  fi
  "$HIPCC" "$@" -c -o "$out" "$src" || exit $?
  echo "plain: hipcc -c ($why)" > "$out.how"
  exit 0
}

[ "${CRN_KEEP_ASM_NOPS:-0}" = "1" ] && build_plain "asked for (CRN_KEEP_ASM_NOPS=1)" "$@"
[ "$ARCH" = "gfx950" ] || build_plain "target $ARCH is not gfx950" "$@"
toolchain=$("$HIPCC" --version 2>/dev/null | grep -m1 'clang version' || true)
if [ -z "${CRN_ASM_RECORD:-}" ] && ! grep -qxF "toolchain $toolchain" "$expected" 2>/dev/null; then
  build_plain "compiler '$toolchain' is not on asm_nops.expected's list" "$@"
fi
LLVM=${ROCM_LLVM_BIN:-$(dirname "$(readlink -f "$HIPCC")")/../lib/llvm/bin}
for tool in clang lld clang-offload-bundler; do
  [ -x "$LLVM/$tool" ] || build_plain "$LLVM/$tool not found" "$@"
done

tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
step() {   # run one hand-made step; on failure fall back to the plain build
  local what=$1; shift
  if ! "$@"; then build_plain "step '$what' failed" "${HIPCC_ARGS[@]}"; fi
}
HIPCC_ARGS=("$@")
step "device code to assembly" "$HIPCC" "$@" -Wno-unused-command-line-argument --cuda-device-only -S -o "$tmp/dev.s" "$src"
report=$(python3 "$here/strip_asm_nops.py" "$tmp/dev.s" "$tmp/dev_f.s") || build_plain "strip_asm_nops.py: $report" "$@"
echo "$(basename "$src"): $report"
if [ -n "${CRN_KEEP_ASM:-}" ]; then cp "$tmp/dev.s" "$CRN_KEEP_ASM/$(basename "$src").dev.s"; cp "$tmp/dev_f.s" "$CRN_KEEP_ASM/$(basename "$src").dev_filtered.s"; fi
dropped=$(sed -n 's/.*dropped \([0-9]*\) .*/\1/p' <<<"$report"); kept=$(sed -n 's/.* kept \([0-9]*\) .*/\1/p' <<<"$report")
want=$(awk -v k="$key" '$1 == "unit" && $2 == k { print $3, $4 }' "$expected")
if [ -n "${CRN_ASM_RECORD:-}" ]; then   # `make expected`: note what this toolchain gives instead of checking it
  echo "unit $key $dropped $kept" >> "$CRN_ASM_RECORD"
elif [ "$want" != "$dropped $kept" ]; then
  build_plain "the filter dropped $dropped / kept $kept wait states, asm_nops.expected records '${want:-nothing}' for $key" "$@"
fi
step "assemble" "$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu="$ARCH" -c "$tmp/dev_f.s" -o "$tmp/dev.o"
step "link the code object" "$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$tmp/dev.out" "$tmp/dev.o"
step "bundle" "$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--"$ARCH" \
  -input=/dev/null -input="$tmp/dev.out" -output="$tmp/dev.hipfb" --compress
step "host side" "$HIPCC" "$@" -Wno-unused-command-line-argument --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$tmp/dev.hipfb" -c -o "$out" "$src"
echo "filtered: dropped $dropped, kept $kept ($toolchain)" > "$out.how"
