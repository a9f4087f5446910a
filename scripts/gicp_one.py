"""configs[1] under GeneralizedIcp, a few registrations (profiling target: scripts/pmc_cmd.sh gicp python scripts/gicp_one.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["METHOD"] = "gicp"
sys.argv = [sys.argv[0], "--one"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_trace.py")).read())
