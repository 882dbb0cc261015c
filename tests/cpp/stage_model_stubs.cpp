// stage_model_stubs.cpp -- TEST INFRASTRUCTURE (CPU staging model, see tests/cpp/hip_shim/hip/hip_runtime.h): the kernel launchers behind
// capi.hip, as stubs. Nothing is computed (results are NOT keyswitch outputs: the model times the host side only). Device time is modelled
// as a sleep of HEXL_MODEL_DEVICE_US_PER_KS microseconds per keyswitch (default 0: an infinitely fast device, i.e. the HOST ceiling) on the
// launching thread -- the runner thread of that device, which in the product waits for its stream at the same place.
#include <chrono>
#include <thread>

#include "../../hexl-fpga_amd/csrc/hexl_internal.hpp"

static void device_time(size_t units, const char* var) {
    static const double us = [var] { const char* e = getenv(var); return e ? atof(e) : 0.0; }();
    if (us > 0) std::this_thread::sleep_for(std::chrono::nanoseconds((long long)(us * 1e3 * (double)units)));
}
int hx_launch_ntt_fwd(hexl_ctx*, u64*, size_t batch, const u64*, const u64*, u64, u64) { device_time(batch, "HEXL_MODEL_DEVICE_US_PER_NTT"); return 0; }
int hx_launch_ntt_inv(hexl_ctx*, u64*, size_t batch, const u64*, const u64*, u64, u64, u64, u64, u64, u64) { device_time(batch, "HEXL_MODEL_DEVICE_US_PER_NTT"); return 0; }
int hx_launch_dyadic(hexl_ctx*, u64*, const u64*, const u64*, size_t, u64, const u64*, u64) { return 0; }
int hx_launch_keyswitch(hexl_ks_plan* p, u64*, const u64*, size_t batch, int, hipEvent_t*) {
    if (!p->have_keys) return HEXL_E_NOKEYS;
    device_time(batch, "HEXL_MODEL_DEVICE_US_PER_KS");
    return 0;
}
int hx_launch_multiply_relinearize(hexl_ks_plan*, u64*, const u64*, const u64*, size_t) { return 0; }
bool hx_ks_can_overwrite(const hexl_ks_plan*, size_t) { return false; }
size_t hx_ks_chunk(const hexl_ks_plan*) { return 256; }
bool hx_ks_lat_applies(const hexl_ks_plan*, size_t) { return false; }     // (the zero-copy lone path needs a device that publishes its limbs)
u32 hx_ks_x_loge() { return 4; }
size_t hexl_ks_scratch_bytes(const hexl_ks_plan*, size_t) { return 0; }
