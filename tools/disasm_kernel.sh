#!/bin/bash
# Disassemble one instantiation of rt_trace_tiles from a device-only code object.
#   tools/disasm_kernel.sh <code-object> <mangled-fragment, e.g. ILi8ELb1ELi0ELb0ELb0E> > out.s
LLVM=/opt/rocm/lib/llvm/bin
SYM=$($LLVM/llvm-readelf -sW "$1" | grep "rt_trace_tiles$2" | grep FUNC | awk '{print $NF}' | head -1)
$LLVM/llvm-objdump -d --no-show-raw-insn --disassemble-symbols="$SYM" "$1"
