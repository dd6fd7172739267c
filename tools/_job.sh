cd $GRAFT_REPO_ROOT/tools/micro
echo "## two processes, destination pair = a source pair"; (./pk_fma_cotenancy alias 3000 & ./pk_fma_cotenancy alias 3000; wait)
echo "## two processes, destination disjoint from the sources"; (./pk_fma_cotenancy plain 3000 & ./pk_fma_cotenancy plain 3000; wait)
echo "## one process alone, aliasing form"; ./pk_fma_cotenancy alias 3000
