import sys, os
lib = os.path.abspath(sys.argv[1])
sys.path.insert(0, '.')
import photonbend_amd.build as b
b.LIB_PATH = lib
import photonbend_amd._native as nat
nat.LIB_PATH = lib
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open('experiments/diag_cu.py').read())
