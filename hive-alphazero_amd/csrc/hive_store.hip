// hive_store.hip -- a content-addressed store of leaf evaluations that outlives a search.
//
// The reference throws its tree away on every move (woker/solo_play.py:103-112: HivePlayer.action starts from an empty
// dict), so the subtree under the move just played -- and every opening position of every later game -- is evaluated
// again.  A leaf's prediction is a pure function of its 448 bytes (HiveBoard + HiveHistory -> planes -> network), and
// since round 4 the engine's (p, v) for a position are the same BITS in any batch and at any row (hive_nn_heads sums over a
// fixed K split; every tower form is bit-identical): a stored (p, v) can stand in for a fresh evaluation without changing
// the search.  The store is an open-addressing table key -> slot over a ring of `capacity` entries in HBM
// (448 bytes of key material + 1584 + 1 floats each); the oldest entries are overwritten; a table entry whose slot has
// been reused is recognised (the slot's key no longer matches) and treated as empty.
//
//   hive_leaf_store_lookup   after hive_leaf_dedup_launch: every row still flagged in `need` whose position is in the store
//                            (64-bit key, then all 448 bytes compared) is switched off; hit[row] = its slot, else -1
//   hive_leaf_store_update   after the forward: row i takes (p, v) of slot hit[rep[i]] if that is a hit; every row that was
//                            evaluated (need still 1) is inserted
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hive_abi.h"

namespace hive {
int set_error(int code, const std::string &msg);
}

struct HiveLeafStore {
    int device = 0;
    int capacity = 0, table = 0;              // entries in the ring; table slots (power of two, 4 x capacity)
    unsigned long long *slot_key = nullptr;   // [capacity] key of the entry in the slot (0 = never used)
    unsigned long long *material = nullptr;   // [capacity][56] the 448 bytes
    float *payload = nullptr;                 // [capacity][1586]: p[1584], v, pad
    unsigned long long *table_key = nullptr;  // [table]
    int *table_slot = nullptr;                // [table] slot or -1
    unsigned long long *counters = nullptr;   // [0] ring position, [1] rows served, [2] rows inserted
};

namespace {
constexpr int kWords = 56, kPay = 1586, kProbes = 48;

#define STORE_TRY(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return hive::set_error(HIVE_E_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

__device__ __forceinline__ unsigned long long row_word(const HiveBoard *boards, const HiveHistory *hist, int row, int lane)
{
    if (lane < 8) return reinterpret_cast<const unsigned long long *>(boards + row)[lane];
    if (lane < kWords) return reinterpret_cast<const unsigned long long *>(hist + row)[lane - 8];
    return 0ull;
}

// one wave per row
__global__ void __launch_bounds__(256)
store_lookup_kernel(const HiveBoard *__restrict__ boards, const HiveHistory *__restrict__ hist, int n,
                    const unsigned long long *__restrict__ keys, int8_t *__restrict__ need, int32_t *__restrict__ hit,
                    unsigned long long *__restrict__ total, int capacity, int table, const unsigned long long *__restrict__ table_key,
                    const int *__restrict__ table_slot, const unsigned long long *__restrict__ slot_key,
                    const unsigned long long *__restrict__ material, unsigned long long *__restrict__ counters)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    int found = -1;
    if (need[row]) {
        const unsigned long long key = keys[row];
        const unsigned long long mine = row_word(boards, hist, row, lane);
        const int ring = (int)(counters[0] % (unsigned long long)capacity);
        for (int pr = 0; pr < kProbes; ++pr) {
            const int idx = (int)((key + (unsigned long long)pr) & (unsigned long long)(table - 1));
            const int s = table_slot[idx];
            if (s < 0) break;                                        // an empty table slot ends the probe sequence
            if (table_key[idx] != key || slot_key[s] != key) continue;   // another key, or a slot that has been reused since
            // (the ring's next n slots may be rewritten by this very batch's update launch while served rows read theirs)
            if ((unsigned long long)((s + capacity - ring) % capacity) < (unsigned long long)n) continue;
            const unsigned long long theirs = lane < kWords ? material[(size_t)s * kWords + lane] : 0ull;
            if (__ballot(mine != theirs) == 0ull) { found = s; break; }
        }
        if (found >= 0 && lane == 0) {
            need[row] = 0;
            atomicAdd(&counters[1], 1ull);
            if (total) atomicAdd(total, ~0ull);                      // one row fewer goes through the tower
        }
    }
    if (lane == 0) hit[row] = found;
}

__global__ void __launch_bounds__(256)
store_update_kernel(const HiveBoard *__restrict__ boards, const HiveHistory *__restrict__ hist, int n,
                    const unsigned long long *__restrict__ keys, const int8_t *__restrict__ need, const int32_t *__restrict__ rep,
                    const int32_t *__restrict__ hit, float *__restrict__ p, float *__restrict__ v, int capacity, int table,
                    unsigned long long *__restrict__ table_key, int *__restrict__ table_slot, unsigned long long *__restrict__ slot_key,
                    unsigned long long *__restrict__ material, float *__restrict__ payload, unsigned long long *__restrict__ counters)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int src = rep ? rep[row] : row;
    const int h = hit[src];
    if (h >= 0) {                                                    // served: this row (or its representative) was in the store
        const float *pay = payload + (size_t)h * kPay;
        for (int k = lane; k < 1584; k += 64) p[(size_t)row * 1584 + k] = pay[k];
        if (lane == 0) v[row] = pay[1584];
        return;
    }
    if (!need[row]) return;                                          // unread, or a duplicate of an evaluated row: nothing to keep
    // evaluated: keep it.  (Two rows of one batch never carry the same position here: hive_leaf_dedup_launch left one.)
    unsigned long long pos = 0ull;
    if (lane == 0) {
        pos = atomicAdd(&counters[0], 1ull);
        atomicAdd(&counters[2], 1ull);
    }
    pos = __shfl(pos, 0, 64);
    const int s = (int)(pos % (unsigned long long)capacity);
    const unsigned long long key = keys[row];
    if (lane < kWords) material[(size_t)s * kWords + lane] = row_word(boards, hist, row, lane);
    float *pay = payload + (size_t)s * kPay;
    for (int k = lane; k < 1584; k += 64) pay[k] = p[(size_t)row * 1584 + k];
    if (lane == 0) {
        pay[1584] = v[row];
        slot_key[s] = key;                                           // (the slot's previous tenant is stale from here on)
        for (int pr = 0; pr < kProbes; ++pr) {
            const int idx = (int)((key + (unsigned long long)pr) & (unsigned long long)(table - 1));
            const int old = table_slot[idx];
            bool free_ = old < 0;
            if (!free_) {
                const unsigned long long tk = table_key[idx];
                free_ = tk == key || slot_key[old] != tk;            // our own older entry, or a stale one
            }
            if (free_ && atomicCAS(&table_slot[idx], old, s) == old) {
                table_key[idx] = key;
                break;
            }
        }                                                            // (a full probe window: the entry is simply not findable)
    }
}

__global__ void store_clear_kernel(int capacity, int table, unsigned long long *slot_key, unsigned long long *table_key, int *table_slot,
                                   unsigned long long *counters)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)capacity) slot_key[i] = 0ull;
    if (i < (size_t)table) { table_key[i] = 0ull; table_slot[i] = -1; }
    if (i < 3) counters[i] = 0ull;
}
}  // namespace

extern "C" int hive_leaf_store_clear(HiveLeafStore *s, void *stream)
{
    if (!s) return hive::set_error(HIVE_E_ARG, "hive_leaf_store_clear: null handle");
    const unsigned blocks = (unsigned)((s->table + 255) / 256);
    hipLaunchKernelGGL(store_clear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->capacity, s->table, s->slot_key,
                       s->table_key, s->table_slot, s->counters);
    STORE_TRY(hipGetLastError());
    return HIVE_OK;
}

extern "C" int hive_leaf_store_destroy(HiveLeafStore *s)
{
    if (!s) return HIVE_OK;
    (void)hipFree(s->slot_key);
    (void)hipFree(s->material);
    (void)hipFree(s->payload);
    (void)hipFree(s->table_key);
    (void)hipFree(s->table_slot);
    (void)hipFree(s->counters);
    delete s;
    return HIVE_OK;
}

extern "C" int hive_leaf_store_create(int device, int capacity, HiveLeafStore **out)
{
    if (!out || capacity < 1024 || capacity > (1 << 22)) return hive::set_error(HIVE_E_ARG, "hive_leaf_store_create: capacity 1024 .. 4 M entries");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return hive::set_error(HIVE_E_DEVICE, "hive_leaf_store_create: no such HIP device");
    STORE_TRY(hipSetDevice(device));
    HiveLeafStore *s = new HiveLeafStore();
    s->device = device;
    s->capacity = capacity;
    s->table = 1;
    while (s->table < 4 * capacity) s->table <<= 1;
    hipError_t e = hipMalloc(&s->slot_key, sizeof(unsigned long long) * (size_t)capacity);
    if (e == hipSuccess) e = hipMalloc(&s->material, sizeof(unsigned long long) * kWords * (size_t)capacity);
    if (e == hipSuccess) e = hipMalloc(&s->payload, sizeof(float) * kPay * (size_t)capacity);
    if (e == hipSuccess) e = hipMalloc(&s->table_key, sizeof(unsigned long long) * (size_t)s->table);
    if (e == hipSuccess) e = hipMalloc(&s->table_slot, sizeof(int) * (size_t)s->table);
    if (e == hipSuccess) e = hipMalloc(&s->counters, sizeof(unsigned long long) * 4);
    if (e != hipSuccess) {
        hive_leaf_store_destroy(s);
        return hive::set_error(HIVE_E_DEVICE, std::string("hive_leaf_store_create: ") + hipGetErrorString(e));
    }
    int rc = hive_leaf_store_clear(s, nullptr);
    if (rc != HIVE_OK) { hive_leaf_store_destroy(s); return rc; }
    STORE_TRY(hipDeviceSynchronize());
    *out = s;
    return HIVE_OK;
}

extern "C" int hive_leaf_store_lookup(HiveLeafStore *s, const HiveBoard *boards, const HiveHistory *hist, int n, const uint64_t *keys,
                                      int8_t *need, int32_t *hit, uint64_t *total, void *stream)
{
    if (!s || !boards || !hist || !keys || !need || !hit || n <= 0) return hive::set_error(HIVE_E_ARG, "hive_leaf_store_lookup: bad argument");
    hipLaunchKernelGGL(store_lookup_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, boards, hist, n,
                       reinterpret_cast<const unsigned long long *>(keys), need, hit, reinterpret_cast<unsigned long long *>(total),
                       s->capacity, s->table, s->table_key, s->table_slot, s->slot_key, s->material, s->counters);
    STORE_TRY(hipGetLastError());
    return HIVE_OK;
}

extern "C" int hive_leaf_store_update(HiveLeafStore *s, const HiveBoard *boards, const HiveHistory *hist, int n, const uint64_t *keys,
                                      const int8_t *need, const int32_t *rep, const int32_t *hit, float *p, float *v, void *stream)
{
    if (!s || !boards || !hist || !keys || !need || !hit || !p || !v || n <= 0)
        return hive::set_error(HIVE_E_ARG, "hive_leaf_store_update: bad argument");
    hipLaunchKernelGGL(store_update_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, boards, hist, n,
                       reinterpret_cast<const unsigned long long *>(keys), need, rep, hit, p, v, s->capacity, s->table, s->table_key,
                       s->table_slot, s->slot_key, s->material, s->payload, s->counters);
    STORE_TRY(hipGetLastError());
    return HIVE_OK;
}

extern "C" int hive_leaf_store_stats(HiveLeafStore *s, uint64_t *served, uint64_t *inserted)
{
    if (!s) return hive::set_error(HIVE_E_ARG, "hive_leaf_store_stats: null handle");
    unsigned long long c[3] = {0, 0, 0};
    STORE_TRY(hipMemcpy(c, s->counters, sizeof(c), hipMemcpyDeviceToHost));
    if (served) *served = c[1];
    if (inserted) *inserted = c[2];
    return HIVE_OK;
}
