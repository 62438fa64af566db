// Wave-cooperative fetch of 64 descriptor rows (272 bytes each, one per lane) through LDS.  gfx950 only, device only.
//
// A lane that fetches its own row needs 17 dwordx4 loads = 17 accesses of the CU's L1 (one 16-byte piece each); with 64
// scattered rows per wave the L1's access rate (about one per cycle) is the bound long before any bandwidth is
// (TCP_TOTAL_CACHE_ACCESSES, profiles/r02_pmc_resolve.txt).  Here one global_load_lds_dwordx4 covers 4 rows x 16 pieces:
// lanes 16s..16s+15 read 256 contiguous bytes of row 4k+s, straight into LDS (the LDS address of such a load is
// wave-uniform base + 16*lane, so instruction k fills LDS rows 4k..4k+3 of 256 bytes).  Column g of LDS row r holds
// piece g ^ (r & 15), so that when afterwards every lane reads piece j of ITS row (ds_read_b128) 16 consecutive lanes
// hit 16 different 16-byte columns: conflict free.  The 17th piece (bytes 256..271) is loaded by the row's own lane
// into a register.  fetch() copies the row to registers, after which the stage buffer is free for the next issue():
// the fetch of round r+1 then overlaps the arithmetic of round r.
//
// ONE WAVE PER WORKGROUP: the __syncthreads() below are wave-level ordering points only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ROW_STAGE_BYTES (64 * 256)
typedef float rs_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char *rs_gptr;

struct RowStage {
    char *stage;            // ROW_STAGE_BYTES of LDS, 1024-byte aligned
    uint32_t *cand;         // 64 words of LDS, 16-byte aligned: the round's row offsets, [lane & 3][lane >> 2]
    uint32_t pz0, rd0;
    int lane, sub;
    rs_f4 tail;

    __device__ __forceinline__ void init(char *stage_, uint32_t *cand_, int lane_)
    {
        stage = stage_; cand = cand_; lane = lane_; sub = lane_ >> 4;
        pz0 = (uint32_t)((lane_ & 15) ^ sub);                       // piece of instruction k: pz0 ^ (4k & 15)
        rd0 = (uint32_t)(lane_ * 256 + ((lane_ & 15) << 4));        // own row: piece j is at rd0 ^ (j << 4)
        tail = (rs_f4){0.f, 0.f, 0.f, 0.f};
    }

    // start fetching: this lane's row begins 16 * off16 bytes after base (inactive lanes fetch nothing)
    __device__ __forceinline__ void issue(rs_gptr base, uint32_t off16, bool act)
    {
        cand[(lane & 3) * 16 + (lane >> 2)] = act ? off16 : 0xFFFFFFFFu;
        __syncthreads();
        const uint4 *sc = reinterpret_cast<const uint4 *>(&cand[sub * 16]);
        uint32_t cr[16];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) { const uint4 v = sc[k4]; cr[4 * k4] = v.x; cr[4 * k4 + 1] = v.y; cr[4 * k4 + 2] = v.z; cr[4 * k4 + 3] = v.w; }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (cr[k] != 0xFFFFFFFFu) {
                const uint32_t piece = pz0 ^ (uint32_t)((4 * k) & 15);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + ((unsigned long long)(cr[k] + piece) << 4)),
                                                 (__attribute__((address_space(3))) void *)(stage + k * 1024), 16, 0, 0);
            }
        }
        if (act) tail = *reinterpret_cast<const __attribute__((address_space(1))) rs_f4 *>(base + ((unsigned long long)(off16 + 16u) << 4));
    }

    // wait for the rows issued last and copy this lane's row to registers
    __device__ __forceinline__ void fetch(float4 (&cv)[17])
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) cv[j] = *reinterpret_cast<const float4 *>(stage + (rd0 ^ (uint32_t)(j << 4)));
        cv[16] = make_float4(tail.x, tail.y, tail.z, tail.w);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
};
