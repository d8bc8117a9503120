// Debug aid (profiles/r06_transposed_gather_attempt.txt): a wave fills v8..v55 (an allocation of 56 VGPRs, or 64 with v63 touched), sleeps while its
// neighbours on the SIMD run -- other copies of itself, or whatever else is on the device (run it beside the kernel under suspicion) -- and counts, per
// register index, the lanes that did not keep their value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(128) void k56(unsigned* __restrict__ hist, unsigned dwell) {
    __shared__ unsigned sm[2500];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned pat = threadIdx.x * 2654435761u + blockIdx.x + sm[(threadIdx.x * 7) & 127], mlo = 0, mhi = 0;
    asm volatile(
        "v_add_u32 v8, 8, %2\n\t"
        "v_add_u32 v9, 9, %2\n\t"
        "v_add_u32 v10, 10, %2\n\t"
        "v_add_u32 v11, 11, %2\n\t"
        "v_add_u32 v12, 12, %2\n\t"
        "v_add_u32 v13, 13, %2\n\t"
        "v_add_u32 v14, 14, %2\n\t"
        "v_add_u32 v15, 15, %2\n\t"
        "v_add_u32 v16, 16, %2\n\t"
        "v_add_u32 v17, 17, %2\n\t"
        "v_add_u32 v18, 18, %2\n\t"
        "v_add_u32 v19, 19, %2\n\t"
        "v_add_u32 v20, 20, %2\n\t"
        "v_add_u32 v21, 21, %2\n\t"
        "v_add_u32 v22, 22, %2\n\t"
        "v_add_u32 v23, 23, %2\n\t"
        "v_add_u32 v24, 24, %2\n\t"
        "v_add_u32 v25, 25, %2\n\t"
        "v_add_u32 v26, 26, %2\n\t"
        "v_add_u32 v27, 27, %2\n\t"
        "v_add_u32 v28, 28, %2\n\t"
        "v_add_u32 v29, 29, %2\n\t"
        "v_add_u32 v30, 30, %2\n\t"
        "v_add_u32 v31, 31, %2\n\t"
        "v_add_u32 v32, 32, %2\n\t"
        "v_add_u32 v33, 33, %2\n\t"
        "v_add_u32 v34, 34, %2\n\t"
        "v_add_u32 v35, 35, %2\n\t"
        "v_add_u32 v36, 36, %2\n\t"
        "v_add_u32 v37, 37, %2\n\t"
        "v_add_u32 v38, 38, %2\n\t"
        "v_add_u32 v39, 39, %2\n\t"
        "v_add_u32 v40, 40, %2\n\t"
        "v_add_u32 v41, 41, %2\n\t"
        "v_add_u32 v42, 42, %2\n\t"
        "v_add_u32 v43, 43, %2\n\t"
        "v_add_u32 v44, 44, %2\n\t"
        "v_add_u32 v45, 45, %2\n\t"
        "v_add_u32 v46, 46, %2\n\t"
        "v_add_u32 v47, 47, %2\n\t"
        "v_add_u32 v48, 48, %2\n\t"
        "v_add_u32 v49, 49, %2\n\t"
        "v_add_u32 v50, 50, %2\n\t"
        "v_add_u32 v51, 51, %2\n\t"
        "v_add_u32 v52, 52, %2\n\t"
        "v_add_u32 v53, 53, %2\n\t"
        "v_add_u32 v54, 54, %2\n\t"
        "v_add_u32 v55, 55, %2\n\t"
        "s_mov_b32 s40, %3\n\t"
        "1:\n\t"
        "s_sleep 20\n\t"
        "s_sub_u32 s40, s40, 1\n\t"
        "s_cmp_lg_u32 s40, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_sub_u32 v8, v8, %2\n\t" "v_cmp_ne_u32 vcc, 8, v8\n\t" "v_cndmask_b32 v8, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v8, 8, %0\n\t"
        "v_sub_u32 v9, v9, %2\n\t" "v_cmp_ne_u32 vcc, 9, v9\n\t" "v_cndmask_b32 v9, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v9, 9, %0\n\t"
        "v_sub_u32 v10, v10, %2\n\t" "v_cmp_ne_u32 vcc, 10, v10\n\t" "v_cndmask_b32 v10, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v10, 10, %0\n\t"
        "v_sub_u32 v11, v11, %2\n\t" "v_cmp_ne_u32 vcc, 11, v11\n\t" "v_cndmask_b32 v11, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v11, 11, %0\n\t"
        "v_sub_u32 v12, v12, %2\n\t" "v_cmp_ne_u32 vcc, 12, v12\n\t" "v_cndmask_b32 v12, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v12, 12, %0\n\t"
        "v_sub_u32 v13, v13, %2\n\t" "v_cmp_ne_u32 vcc, 13, v13\n\t" "v_cndmask_b32 v13, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v13, 13, %0\n\t"
        "v_sub_u32 v14, v14, %2\n\t" "v_cmp_ne_u32 vcc, 14, v14\n\t" "v_cndmask_b32 v14, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v14, 14, %0\n\t"
        "v_sub_u32 v15, v15, %2\n\t" "v_cmp_ne_u32 vcc, 15, v15\n\t" "v_cndmask_b32 v15, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v15, 15, %0\n\t"
        "v_sub_u32 v16, v16, %2\n\t" "v_cmp_ne_u32 vcc, 16, v16\n\t" "v_cndmask_b32 v16, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v16, 16, %0\n\t"
        "v_sub_u32 v17, v17, %2\n\t" "v_cmp_ne_u32 vcc, 17, v17\n\t" "v_cndmask_b32 v17, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v17, 17, %0\n\t"
        "v_sub_u32 v18, v18, %2\n\t" "v_cmp_ne_u32 vcc, 18, v18\n\t" "v_cndmask_b32 v18, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v18, 18, %0\n\t"
        "v_sub_u32 v19, v19, %2\n\t" "v_cmp_ne_u32 vcc, 19, v19\n\t" "v_cndmask_b32 v19, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v19, 19, %0\n\t"
        "v_sub_u32 v20, v20, %2\n\t" "v_cmp_ne_u32 vcc, 20, v20\n\t" "v_cndmask_b32 v20, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v20, 20, %0\n\t"
        "v_sub_u32 v21, v21, %2\n\t" "v_cmp_ne_u32 vcc, 21, v21\n\t" "v_cndmask_b32 v21, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v21, 21, %0\n\t"
        "v_sub_u32 v22, v22, %2\n\t" "v_cmp_ne_u32 vcc, 22, v22\n\t" "v_cndmask_b32 v22, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v22, 22, %0\n\t"
        "v_sub_u32 v23, v23, %2\n\t" "v_cmp_ne_u32 vcc, 23, v23\n\t" "v_cndmask_b32 v23, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v23, 23, %0\n\t"
        "v_sub_u32 v24, v24, %2\n\t" "v_cmp_ne_u32 vcc, 24, v24\n\t" "v_cndmask_b32 v24, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v24, 24, %0\n\t"
        "v_sub_u32 v25, v25, %2\n\t" "v_cmp_ne_u32 vcc, 25, v25\n\t" "v_cndmask_b32 v25, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v25, 25, %0\n\t"
        "v_sub_u32 v26, v26, %2\n\t" "v_cmp_ne_u32 vcc, 26, v26\n\t" "v_cndmask_b32 v26, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v26, 26, %0\n\t"
        "v_sub_u32 v27, v27, %2\n\t" "v_cmp_ne_u32 vcc, 27, v27\n\t" "v_cndmask_b32 v27, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v27, 27, %0\n\t"
        "v_sub_u32 v28, v28, %2\n\t" "v_cmp_ne_u32 vcc, 28, v28\n\t" "v_cndmask_b32 v28, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v28, 28, %0\n\t"
        "v_sub_u32 v29, v29, %2\n\t" "v_cmp_ne_u32 vcc, 29, v29\n\t" "v_cndmask_b32 v29, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v29, 29, %0\n\t"
        "v_sub_u32 v30, v30, %2\n\t" "v_cmp_ne_u32 vcc, 30, v30\n\t" "v_cndmask_b32 v30, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v30, 30, %0\n\t"
        "v_sub_u32 v31, v31, %2\n\t" "v_cmp_ne_u32 vcc, 31, v31\n\t" "v_cndmask_b32 v31, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v31, 31, %0\n\t"
        "v_sub_u32 v32, v32, %2\n\t" "v_cmp_ne_u32 vcc, 32, v32\n\t" "v_cndmask_b32 v32, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v32, 0, %1\n\t"
        "v_sub_u32 v33, v33, %2\n\t" "v_cmp_ne_u32 vcc, 33, v33\n\t" "v_cndmask_b32 v33, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v33, 1, %1\n\t"
        "v_sub_u32 v34, v34, %2\n\t" "v_cmp_ne_u32 vcc, 34, v34\n\t" "v_cndmask_b32 v34, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v34, 2, %1\n\t"
        "v_sub_u32 v35, v35, %2\n\t" "v_cmp_ne_u32 vcc, 35, v35\n\t" "v_cndmask_b32 v35, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v35, 3, %1\n\t"
        "v_sub_u32 v36, v36, %2\n\t" "v_cmp_ne_u32 vcc, 36, v36\n\t" "v_cndmask_b32 v36, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v36, 4, %1\n\t"
        "v_sub_u32 v37, v37, %2\n\t" "v_cmp_ne_u32 vcc, 37, v37\n\t" "v_cndmask_b32 v37, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v37, 5, %1\n\t"
        "v_sub_u32 v38, v38, %2\n\t" "v_cmp_ne_u32 vcc, 38, v38\n\t" "v_cndmask_b32 v38, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v38, 6, %1\n\t"
        "v_sub_u32 v39, v39, %2\n\t" "v_cmp_ne_u32 vcc, 39, v39\n\t" "v_cndmask_b32 v39, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v39, 7, %1\n\t"
        "v_sub_u32 v40, v40, %2\n\t" "v_cmp_ne_u32 vcc, 40, v40\n\t" "v_cndmask_b32 v40, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v40, 8, %1\n\t"
        "v_sub_u32 v41, v41, %2\n\t" "v_cmp_ne_u32 vcc, 41, v41\n\t" "v_cndmask_b32 v41, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v41, 9, %1\n\t"
        "v_sub_u32 v42, v42, %2\n\t" "v_cmp_ne_u32 vcc, 42, v42\n\t" "v_cndmask_b32 v42, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v42, 10, %1\n\t"
        "v_sub_u32 v43, v43, %2\n\t" "v_cmp_ne_u32 vcc, 43, v43\n\t" "v_cndmask_b32 v43, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v43, 11, %1\n\t"
        "v_sub_u32 v44, v44, %2\n\t" "v_cmp_ne_u32 vcc, 44, v44\n\t" "v_cndmask_b32 v44, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v44, 12, %1\n\t"
        "v_sub_u32 v45, v45, %2\n\t" "v_cmp_ne_u32 vcc, 45, v45\n\t" "v_cndmask_b32 v45, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v45, 13, %1\n\t"
        "v_sub_u32 v46, v46, %2\n\t" "v_cmp_ne_u32 vcc, 46, v46\n\t" "v_cndmask_b32 v46, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v46, 14, %1\n\t"
        "v_sub_u32 v47, v47, %2\n\t" "v_cmp_ne_u32 vcc, 47, v47\n\t" "v_cndmask_b32 v47, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v47, 15, %1\n\t"
        "v_sub_u32 v48, v48, %2\n\t" "v_cmp_ne_u32 vcc, 48, v48\n\t" "v_cndmask_b32 v48, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v48, 16, %1\n\t"
        "v_sub_u32 v49, v49, %2\n\t" "v_cmp_ne_u32 vcc, 49, v49\n\t" "v_cndmask_b32 v49, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v49, 17, %1\n\t"
        "v_sub_u32 v50, v50, %2\n\t" "v_cmp_ne_u32 vcc, 50, v50\n\t" "v_cndmask_b32 v50, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v50, 18, %1\n\t"
        "v_sub_u32 v51, v51, %2\n\t" "v_cmp_ne_u32 vcc, 51, v51\n\t" "v_cndmask_b32 v51, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v51, 19, %1\n\t"
        "v_sub_u32 v52, v52, %2\n\t" "v_cmp_ne_u32 vcc, 52, v52\n\t" "v_cndmask_b32 v52, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v52, 20, %1\n\t"
        "v_sub_u32 v53, v53, %2\n\t" "v_cmp_ne_u32 vcc, 53, v53\n\t" "v_cndmask_b32 v53, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v53, 21, %1\n\t"
        "v_sub_u32 v54, v54, %2\n\t" "v_cmp_ne_u32 vcc, 54, v54\n\t" "v_cndmask_b32 v54, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v54, 22, %1\n\t"
        "v_sub_u32 v55, v55, %2\n\t" "v_cmp_ne_u32 vcc, 55, v55\n\t" "v_cndmask_b32 v55, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v55, 23, %1\n\t"
        : "+v"(mlo), "+v"(mhi) : "v"(pat), "s"(dwell) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "vcc", "scc", "s40");
    for (unsigned r = 0; r < 64; ++r)
        if (((r < 32 ? mlo : mhi) >> (r & 31)) & 1u) atomicAdd(&hist[r], 1u);
}

__global__ __launch_bounds__(128) void k64(unsigned* __restrict__ hist, unsigned dwell) {
    __shared__ unsigned sm[2500];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned pat = threadIdx.x * 2654435761u + blockIdx.x + sm[(threadIdx.x * 7) & 127], mlo = 0, mhi = 0;
    asm volatile(
        "v_add_u32 v8, 8, %2\n\t"
        "v_add_u32 v9, 9, %2\n\t"
        "v_add_u32 v10, 10, %2\n\t"
        "v_add_u32 v11, 11, %2\n\t"
        "v_add_u32 v12, 12, %2\n\t"
        "v_add_u32 v13, 13, %2\n\t"
        "v_add_u32 v14, 14, %2\n\t"
        "v_add_u32 v15, 15, %2\n\t"
        "v_add_u32 v16, 16, %2\n\t"
        "v_add_u32 v17, 17, %2\n\t"
        "v_add_u32 v18, 18, %2\n\t"
        "v_add_u32 v19, 19, %2\n\t"
        "v_add_u32 v20, 20, %2\n\t"
        "v_add_u32 v21, 21, %2\n\t"
        "v_add_u32 v22, 22, %2\n\t"
        "v_add_u32 v23, 23, %2\n\t"
        "v_add_u32 v24, 24, %2\n\t"
        "v_add_u32 v25, 25, %2\n\t"
        "v_add_u32 v26, 26, %2\n\t"
        "v_add_u32 v27, 27, %2\n\t"
        "v_add_u32 v28, 28, %2\n\t"
        "v_add_u32 v29, 29, %2\n\t"
        "v_add_u32 v30, 30, %2\n\t"
        "v_add_u32 v31, 31, %2\n\t"
        "v_add_u32 v32, 32, %2\n\t"
        "v_add_u32 v33, 33, %2\n\t"
        "v_add_u32 v34, 34, %2\n\t"
        "v_add_u32 v35, 35, %2\n\t"
        "v_add_u32 v36, 36, %2\n\t"
        "v_add_u32 v37, 37, %2\n\t"
        "v_add_u32 v38, 38, %2\n\t"
        "v_add_u32 v39, 39, %2\n\t"
        "v_add_u32 v40, 40, %2\n\t"
        "v_add_u32 v41, 41, %2\n\t"
        "v_add_u32 v42, 42, %2\n\t"
        "v_add_u32 v43, 43, %2\n\t"
        "v_add_u32 v44, 44, %2\n\t"
        "v_add_u32 v45, 45, %2\n\t"
        "v_add_u32 v46, 46, %2\n\t"
        "v_add_u32 v47, 47, %2\n\t"
        "v_add_u32 v48, 48, %2\n\t"
        "v_add_u32 v49, 49, %2\n\t"
        "v_add_u32 v50, 50, %2\n\t"
        "v_add_u32 v51, 51, %2\n\t"
        "v_add_u32 v52, 52, %2\n\t"
        "v_add_u32 v53, 53, %2\n\t"
        "v_add_u32 v54, 54, %2\n\t"
        "v_add_u32 v55, 55, %2\n\t"
        "s_mov_b32 s40, %3\n\t"
        "1:\n\t"
        "s_sleep 20\n\t"
        "s_sub_u32 s40, s40, 1\n\t"
        "s_cmp_lg_u32 s40, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_sub_u32 v8, v8, %2\n\t" "v_cmp_ne_u32 vcc, 8, v8\n\t" "v_cndmask_b32 v8, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v8, 8, %0\n\t"
        "v_sub_u32 v9, v9, %2\n\t" "v_cmp_ne_u32 vcc, 9, v9\n\t" "v_cndmask_b32 v9, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v9, 9, %0\n\t"
        "v_sub_u32 v10, v10, %2\n\t" "v_cmp_ne_u32 vcc, 10, v10\n\t" "v_cndmask_b32 v10, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v10, 10, %0\n\t"
        "v_sub_u32 v11, v11, %2\n\t" "v_cmp_ne_u32 vcc, 11, v11\n\t" "v_cndmask_b32 v11, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v11, 11, %0\n\t"
        "v_sub_u32 v12, v12, %2\n\t" "v_cmp_ne_u32 vcc, 12, v12\n\t" "v_cndmask_b32 v12, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v12, 12, %0\n\t"
        "v_sub_u32 v13, v13, %2\n\t" "v_cmp_ne_u32 vcc, 13, v13\n\t" "v_cndmask_b32 v13, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v13, 13, %0\n\t"
        "v_sub_u32 v14, v14, %2\n\t" "v_cmp_ne_u32 vcc, 14, v14\n\t" "v_cndmask_b32 v14, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v14, 14, %0\n\t"
        "v_sub_u32 v15, v15, %2\n\t" "v_cmp_ne_u32 vcc, 15, v15\n\t" "v_cndmask_b32 v15, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v15, 15, %0\n\t"
        "v_sub_u32 v16, v16, %2\n\t" "v_cmp_ne_u32 vcc, 16, v16\n\t" "v_cndmask_b32 v16, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v16, 16, %0\n\t"
        "v_sub_u32 v17, v17, %2\n\t" "v_cmp_ne_u32 vcc, 17, v17\n\t" "v_cndmask_b32 v17, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v17, 17, %0\n\t"
        "v_sub_u32 v18, v18, %2\n\t" "v_cmp_ne_u32 vcc, 18, v18\n\t" "v_cndmask_b32 v18, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v18, 18, %0\n\t"
        "v_sub_u32 v19, v19, %2\n\t" "v_cmp_ne_u32 vcc, 19, v19\n\t" "v_cndmask_b32 v19, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v19, 19, %0\n\t"
        "v_sub_u32 v20, v20, %2\n\t" "v_cmp_ne_u32 vcc, 20, v20\n\t" "v_cndmask_b32 v20, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v20, 20, %0\n\t"
        "v_sub_u32 v21, v21, %2\n\t" "v_cmp_ne_u32 vcc, 21, v21\n\t" "v_cndmask_b32 v21, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v21, 21, %0\n\t"
        "v_sub_u32 v22, v22, %2\n\t" "v_cmp_ne_u32 vcc, 22, v22\n\t" "v_cndmask_b32 v22, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v22, 22, %0\n\t"
        "v_sub_u32 v23, v23, %2\n\t" "v_cmp_ne_u32 vcc, 23, v23\n\t" "v_cndmask_b32 v23, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v23, 23, %0\n\t"
        "v_sub_u32 v24, v24, %2\n\t" "v_cmp_ne_u32 vcc, 24, v24\n\t" "v_cndmask_b32 v24, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v24, 24, %0\n\t"
        "v_sub_u32 v25, v25, %2\n\t" "v_cmp_ne_u32 vcc, 25, v25\n\t" "v_cndmask_b32 v25, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v25, 25, %0\n\t"
        "v_sub_u32 v26, v26, %2\n\t" "v_cmp_ne_u32 vcc, 26, v26\n\t" "v_cndmask_b32 v26, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v26, 26, %0\n\t"
        "v_sub_u32 v27, v27, %2\n\t" "v_cmp_ne_u32 vcc, 27, v27\n\t" "v_cndmask_b32 v27, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v27, 27, %0\n\t"
        "v_sub_u32 v28, v28, %2\n\t" "v_cmp_ne_u32 vcc, 28, v28\n\t" "v_cndmask_b32 v28, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v28, 28, %0\n\t"
        "v_sub_u32 v29, v29, %2\n\t" "v_cmp_ne_u32 vcc, 29, v29\n\t" "v_cndmask_b32 v29, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v29, 29, %0\n\t"
        "v_sub_u32 v30, v30, %2\n\t" "v_cmp_ne_u32 vcc, 30, v30\n\t" "v_cndmask_b32 v30, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v30, 30, %0\n\t"
        "v_sub_u32 v31, v31, %2\n\t" "v_cmp_ne_u32 vcc, 31, v31\n\t" "v_cndmask_b32 v31, 0, 1, vcc\n\t" "v_lshl_or_b32 %0, v31, 31, %0\n\t"
        "v_sub_u32 v32, v32, %2\n\t" "v_cmp_ne_u32 vcc, 32, v32\n\t" "v_cndmask_b32 v32, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v32, 0, %1\n\t"
        "v_sub_u32 v33, v33, %2\n\t" "v_cmp_ne_u32 vcc, 33, v33\n\t" "v_cndmask_b32 v33, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v33, 1, %1\n\t"
        "v_sub_u32 v34, v34, %2\n\t" "v_cmp_ne_u32 vcc, 34, v34\n\t" "v_cndmask_b32 v34, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v34, 2, %1\n\t"
        "v_sub_u32 v35, v35, %2\n\t" "v_cmp_ne_u32 vcc, 35, v35\n\t" "v_cndmask_b32 v35, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v35, 3, %1\n\t"
        "v_sub_u32 v36, v36, %2\n\t" "v_cmp_ne_u32 vcc, 36, v36\n\t" "v_cndmask_b32 v36, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v36, 4, %1\n\t"
        "v_sub_u32 v37, v37, %2\n\t" "v_cmp_ne_u32 vcc, 37, v37\n\t" "v_cndmask_b32 v37, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v37, 5, %1\n\t"
        "v_sub_u32 v38, v38, %2\n\t" "v_cmp_ne_u32 vcc, 38, v38\n\t" "v_cndmask_b32 v38, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v38, 6, %1\n\t"
        "v_sub_u32 v39, v39, %2\n\t" "v_cmp_ne_u32 vcc, 39, v39\n\t" "v_cndmask_b32 v39, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v39, 7, %1\n\t"
        "v_sub_u32 v40, v40, %2\n\t" "v_cmp_ne_u32 vcc, 40, v40\n\t" "v_cndmask_b32 v40, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v40, 8, %1\n\t"
        "v_sub_u32 v41, v41, %2\n\t" "v_cmp_ne_u32 vcc, 41, v41\n\t" "v_cndmask_b32 v41, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v41, 9, %1\n\t"
        "v_sub_u32 v42, v42, %2\n\t" "v_cmp_ne_u32 vcc, 42, v42\n\t" "v_cndmask_b32 v42, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v42, 10, %1\n\t"
        "v_sub_u32 v43, v43, %2\n\t" "v_cmp_ne_u32 vcc, 43, v43\n\t" "v_cndmask_b32 v43, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v43, 11, %1\n\t"
        "v_sub_u32 v44, v44, %2\n\t" "v_cmp_ne_u32 vcc, 44, v44\n\t" "v_cndmask_b32 v44, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v44, 12, %1\n\t"
        "v_sub_u32 v45, v45, %2\n\t" "v_cmp_ne_u32 vcc, 45, v45\n\t" "v_cndmask_b32 v45, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v45, 13, %1\n\t"
        "v_sub_u32 v46, v46, %2\n\t" "v_cmp_ne_u32 vcc, 46, v46\n\t" "v_cndmask_b32 v46, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v46, 14, %1\n\t"
        "v_sub_u32 v47, v47, %2\n\t" "v_cmp_ne_u32 vcc, 47, v47\n\t" "v_cndmask_b32 v47, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v47, 15, %1\n\t"
        "v_sub_u32 v48, v48, %2\n\t" "v_cmp_ne_u32 vcc, 48, v48\n\t" "v_cndmask_b32 v48, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v48, 16, %1\n\t"
        "v_sub_u32 v49, v49, %2\n\t" "v_cmp_ne_u32 vcc, 49, v49\n\t" "v_cndmask_b32 v49, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v49, 17, %1\n\t"
        "v_sub_u32 v50, v50, %2\n\t" "v_cmp_ne_u32 vcc, 50, v50\n\t" "v_cndmask_b32 v50, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v50, 18, %1\n\t"
        "v_sub_u32 v51, v51, %2\n\t" "v_cmp_ne_u32 vcc, 51, v51\n\t" "v_cndmask_b32 v51, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v51, 19, %1\n\t"
        "v_sub_u32 v52, v52, %2\n\t" "v_cmp_ne_u32 vcc, 52, v52\n\t" "v_cndmask_b32 v52, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v52, 20, %1\n\t"
        "v_sub_u32 v53, v53, %2\n\t" "v_cmp_ne_u32 vcc, 53, v53\n\t" "v_cndmask_b32 v53, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v53, 21, %1\n\t"
        "v_sub_u32 v54, v54, %2\n\t" "v_cmp_ne_u32 vcc, 54, v54\n\t" "v_cndmask_b32 v54, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v54, 22, %1\n\t"
        "v_sub_u32 v55, v55, %2\n\t" "v_cmp_ne_u32 vcc, 55, v55\n\t" "v_cndmask_b32 v55, 0, 1, vcc\n\t" "v_lshl_or_b32 %1, v55, 23, %1\n\t"
        : "+v"(mlo), "+v"(mhi) : "v"(pat), "s"(dwell) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v63", "vcc", "scc", "s40");
    for (unsigned r = 0; r < 64; ++r)
        if (((r < 32 ? mlo : mhi) >> (r & 31)) & 1u) atomicAdd(&hist[r], 1u);
}

template <typename K> void run(const char* name, K k, unsigned* hist, int reps, unsigned dwell) {
    unsigned total[64] = {0};
    for (int rep = 0; rep < reps; ++rep) {
        (void)hipMemset(hist, 0, 64 * 4);
        hipLaunchKernelGGL(k, dim3(20000), dim3(128), 0, 0, hist, dwell);
        (void)hipDeviceSynchronize();
        unsigned h[64];
        (void)hipMemcpy(h, hist, 64 * 4, hipMemcpyDeviceToHost);
        for (int r = 0; r < 64; ++r) total[r] += h[r];
    }
    printf("%s, %d launches:", name, reps);
    bool any = false;
    for (int r = 0; r < 64; ++r)
        if (total[r]) { printf(" v%d:%u", r, total[r]); any = true; }
    printf(any ? "\n" : " no register lost its value\n");
    fflush(stdout);
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const unsigned dwell = argc > 2 ? (unsigned)atoi(argv[2]) : 40u;
    unsigned* hist;
    (void)hipMalloc(&hist, 64 * 4);
    run("allocation 56 (v8..v55 held)", k56, hist, reps, dwell);
    run("allocation 64 (v8..v55 held, v63 touched)", k64, hist, reps, dwell);
    return 0;
}

// the same kernels from another program (tools/debug_corun_gpu.py: the probe on one stream, the kernel under suspicion on another, in ONE process so that their waves share SIMDs)
extern "C" int probe_launch(void* stream, unsigned* d_hist, unsigned blocks, unsigned dwell, int alloc64) {
    if (alloc64) hipLaunchKernelGGL(k64, dim3(blocks), dim3(128), 0, (hipStream_t)stream, d_hist, dwell);
    else hipLaunchKernelGGL(k56, dim3(blocks), dim3(128), 0, (hipStream_t)stream, d_hist, dwell);
    return (int)hipGetLastError();
}
