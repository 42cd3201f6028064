cd $GRAFT_REPO_ROOT
for L in libimm3_ablate.so libimm3_p32.so; do
export IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/$L
echo == $L
python tools/sp_explore.py C3 0 66 51 57 2>&1 | tee -a gpurun_out/r4_e3.log
done
