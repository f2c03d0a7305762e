cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in autoregressive "jacobi greedy"; do
timeout 900 python -c "
import cProfile, pstats, sys
sys.argv=['engine_throughput.py','--max-tokens','192','--only','$mode']
sys.path.insert(0,'tools')
import engine_throughput as e
cProfile.run('e.main()','/tmp/eng.prof')
p=pstats.Stats('/tmp/eng.prof'); p.sort_stats('tottime').print_stats(40)
" > "gpurun_out/r2_eng_cprofile_${mode// /_}.log" 2>&1
done
