// ONE launch for all parameter-side preparation of a training step: after the optimizer has changed the weights, every convolution's
// MFMA operand image (implicit-GEMM packing and Winograd filter transform, forward and data-gradient forms), the encoder stem folded
// into the first 3x3 convolution and the flipped filter of the single-output-channel convolution's data gradient are rebuilt by the
// blocks of one grid (the job table travels by value as a kernel argument: graph-capture safe).  At a small data-parallel shard a
// step is ~80 graph nodes of which the four separate preparation launches (wino_pack_many for each network, pack_many,
// thin_stem_fold) and thin_cout1_flip were five, each a ~5 us node for a few KB of work (profiles/r04_small_shard_budget.txt).
// Bodies: aesr_pack_dev.h (the stand-alone kernels run the same code; results are bitwise identical).
#include "aesr_kernels.h"
#include "aesr_pack_dev.h"

__global__ __launch_bounds__(256) void prep_many_kernel(PrepTable t) {
    int j = 0;
    for (int k = 1; k < t.njobs; ++k)
        if ((int)blockIdx.x >= t.job[k].block0) j = k;
    const PrepJob& jb = t.job[j];
    const int b1 = (j + 1 < t.njobs) ? t.job[j + 1].block0 : t.nblocks;
    const size_t first = (size_t)(blockIdx.x - jb.block0) * 256 + threadIdx.x, stride = (size_t)(b1 - jb.block0) * 256;
    switch (jb.kind) {
    case PREP_PACK:
        pack_elements(jb.w, jb.out, jb.Cout, jb.Cin, jb.KS, jb.KinP, jb.NoutP, jb.TN, jb.transpose, first, stride);
        break;
    case PREP_WINO_PACK:
        wino_pack_elements(jb.w, jb.out, jb.Cout, jb.Cin, jb.KinP, jb.NoutP, jb.transpose & 1, first, stride);
        break;
    case PREP_STEM_FOLD:          // w = W1 [C1][Cs][3][3], aux0 = stem weight [Cs], aux1 = stem bias [Cs] or null; Cout = C1, Cin = Cs
        stem_fold_elements(jb.aux0, jb.aux1, jb.w, jb.out, jb.Cin, jb.Cout, (int)first, (int)stride);
        break;
    case PREP_COUT1_FLIP:
        cout1_flip_elements(jb.w, jb.out, jb.Cin, (int)first, (int)stride);
        break;
    default:
        break;
    }
}

int aesr_launch_prep_many(const PrepTable& t, hipStream_t st) {
    hipLaunchKernelGGL(prep_many_kernel, dim3(t.nblocks), dim3(256), 0, st, t);
    AESR_LAUNCH_CHECK("prep_many");
    return AESR_OK;
}
