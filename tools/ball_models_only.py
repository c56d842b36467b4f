import sys, os
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.environ['PYTHONPATH'], 'tools', 'ball_bench.py')).read().replace("for radius in (0.05, 0.1, 0.2):", "for radius in ():"))
