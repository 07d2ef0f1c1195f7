"""cProfile of the host side of the contrastive-pretraining iteration (the step is host-bound at batch 2048)."""
import cProfile, pstats, sys, os, io
sys.argv = [sys.argv[0], "--steps", "20", "--warmup", "3"]
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pretrain_bench.py")).read()
pr.enable()
exec(compile(src, "pretrain_bench.py", "exec"), {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "pretrain_bench.py")})
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
