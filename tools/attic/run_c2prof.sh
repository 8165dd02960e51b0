set -e
python -c "
import cProfile,pstats,sys,io
sys.argv=['bench.py','--single','--no-cpu-baseline','--workload','text8_d64','--batch-size','131072','--steps','900','--warmup','90']
import runpy
pr=cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py',run_name='__main__')
except SystemExit:
    pass
pr.disable()
s=io.StringIO()
pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(35)
open('gpurun_out/c2_cprofile.txt','w').write(s.getvalue())
" > gpurun_out/c2prof.json 2> gpurun_out/c2prof.err
python bench.py --single --no-cpu-baseline --workload text8_d64 --batch-size 131072 --steps 900 --warmup 90 > gpurun_out/c2_dealt.json 2> gpurun_out/c2_dealt.err
python bench.py --single --no-cpu-baseline --workload text8_d64 --batch-size 131072 --steps 900 --warmup 90 --static-index > gpurun_out/c2_static.json 2> gpurun_out/c2_static.err
