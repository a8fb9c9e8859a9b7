# Development: where a kernel of the built library spills.  usage: tools/kspill.sh <code object index> <mangled-name substring>
# prints the line numbers of scratch instructions and of s_barrier in the kernel's disassembly (copy in /tmp/k_last.s)
LIB=/root/repo/project-resnet-bk-an-o-n-language-model-architecture_amd/bkcore/libbkcore.so
D=$(mktemp -d); cp $LIB $D/; cd $D
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading libbkcore.so > /dev/null
CO=$(ls | grep "libbkcore.so.$1.hipv4")
SYM=$(/opt/rocm/lib/llvm/bin/llvm-readelf -sW $CO | awk '{print $8}' | grep -v "\.kd$" | grep "$2" | head -1)
echo $SYM
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$SYM $CO > k.s
wc -l k.s
grep -n "scratch_\|s_barrier" k.s | awk '{print $1, $2}' | tr '\n' ' '
echo
cp k.s /tmp/k_last.s
rm -rf $D
