// Cooperative (one persistent launch per direction) execution of a small MLP chain: declarations shared by mlp.hip (which
// decides per call) and chain_coop.hip (kernels + launchers).  See include/pn2_hip.h "Cooperative chain launches".
#pragma once
#include "pn2_common.h"

namespace pn2 {
namespace coop {

constexpr int kMaxLayers = 4;     // layers per chain
constexpr int kMaxC = 512;        // widest BatchNorm layer whose coefficient block fits the LDS copy

struct FwdCall {
    const float* x;
    int64_t ldx;
    int rows;
    const pn2_mlp_layer* layers;
    int nlayers, pool_k;
    float* out;
    int32_t* arg;
    float* partial[2];            // two statistics-partial buffers (consecutive layers alternate)
    const pn2_coop* ctl;
    hipStream_t stream;
};

struct BwdCall {
    const float* x;
    int64_t ldx;
    int rows;
    const pn2_mlp_layer* layers;
    int nlayers, pool_k;
    const float* dout;
    const int32_t* arg;
    float* dx;
    int64_t lddx;
    int dx_first_col, zero_lead;
    float* scratch[2];
    float* partial[2];
    // weight-gradient slabs: per layer (arena pointer, k_per_split, number of splits); the caller owns the reduction
    float* slab[kMaxLayers];
    int kps[kMaxLayers], nsplit[kMaxLayers];
    const pn2_coop* ctl;
    hipStream_t stream;
};

// Can this call run cooperatively?  (shape limits only; the caller has checked mode, segments and linkage)
bool shapes_ok(int rows, const pn2_mlp_layer* layers, int nlayers, int pool_k);
int forward(const FwdCall& c);
int backward(const BwdCall& c);
// reduction ranges of the weight gradient dW [cout][cin] over `rows` rows for the cooperative backward
void plan_slabs(int rows, int cout, int cin, int* kps, int* nsplit);

}  // namespace coop
}  // namespace pn2
