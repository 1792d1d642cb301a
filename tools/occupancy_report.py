import ctypes as C, sys
sys.path.insert(0,'/root/repo')
from mpvss_rs_amd import capi
lib=capi.load_library()
e=capi.Engine(0)
out=(C.c_int*5)()
lib.modp_occupancy_report(out)
print("resident waves/CU by runtime occupancy API [commit_eval,dual_exp,build_table,to_mont,mul]:", list(out))
