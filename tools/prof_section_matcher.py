"""cProfile of tools/bench_section_matcher.py's second call (host-side cost of one alignment pair)"""
import cProfile
import io
import pstats
import runpy
import sys

sys.argv = ['bench_section_matcher.py'] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
runpy.run_path('tools/bench_section_matcher.py', run_name='__main__')
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:9000])
