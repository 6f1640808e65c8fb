import sys, time, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
csim = load_package(); csim.lib(); csim.set_device(0)
nx, ny = 4096, 8192
d = csim.decomp_init(1, 0, nx, ny)
for k in range(4): d.nbr[k] = 0
st = csim.Stepper(d, 1.0, 1.0, csim.bc_codes("dddd"))
st.comm_init(csim.comm_unique_id())
st.init_gaussian()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    st.run(0.05, 0.1, 0.5, 0.25, 60); st.sync()
st.tune(0.05, 0.1, 0.5, 0.25)
for rep in range(6):
    st.sync()
    st.keep_warm(0.05, 0.1, 0.5, 0.25, 0.002)
    time.sleep(0.0003)
    a = time.perf_counter()
    st.run(0.05, 0.1, 0.5, 0.25, 20)
    b = time.perf_counter()
    st.sync()
    c = time.perf_counter()
    print(f"isolated 20-step call: enqueue {1e6*(b-a):.0f} us, total {1e6*(c-a):.0f} us -> {nx*ny*20/(c-a)/1e6:.0f} Mcell/s")
a = time.perf_counter()
for _ in range(30): st.run(0.05, 0.1, 0.5, 0.25, 20)
b = time.perf_counter(); st.sync(); c = time.perf_counter()
print(f"30 back-to-back calls: enqueue {1e6*(b-a)/30:.0f} us per call, total {1e6*(c-a)/30:.0f} us per call")
