// accbench.hip -- the accumulation kernel of g1_kernels.hip on its own, back to back (hot clocks, no engine around it):
// k_g1_accumulate over a synthetic plan -- 2048 groups x SIZE members, members a random permutation of the table's rows --
// at 8, 16 and 32 members per lane; then k_g1_tree over its lane partials.  Rows are random field elements, not curve
// points: the formulas do not care.  (Round 4 ran the deleted 12 x 32-bit kernel beside it with this tool: 183 vs 158 us at
// 1 M points, lane partials identical word for word -- profiles/r04_accbench.txt.)
// usage: accbench [SIZE = 512] [with_bits = 0]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -I../include -o accbench accbench.hip
#include "../pos_evolution_amd/csrc/g1_kernels.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using namespace posevo;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const uint32_t NG = 2048, SIZE = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
    const uint64_t NV = (uint64_t)NG * SIZE;
    std::mt19937_64 rng(4);
    std::vector<uint32_t> pts((size_t)NV * G1_ROW_WORDS, 0), members(NV);
    for (uint64_t i = 0; i < NV; ++i) {
        for (int k = 0; k < 24; ++k) pts[i * G1_ROW_WORDS + k] = (uint32_t)rng();
        pts[i * G1_ROW_WORDS + 11] &= 0x0fffffffu;
        pts[i * G1_ROW_WORDS + 23] &= 0x0fffffffu;
        members[i] = (uint32_t)i;
    }
    std::shuffle(members.begin(), members.end(), rng);
    uint32_t *d_pts, *d_pts29, *d_mem, *d_lane, *d_lane2, *d_wg, *d_bits;
    const int with_bits = argc > 2 ? atoi(argv[2]) : 0;   // 1: every group has a bitfield (99 % of the bits set)
    {
        std::vector<uint32_t> bits((size_t)NG * ((SIZE + 31) / 32));
        for (auto& w : bits) { w = 0xFFFFFFFFu; if ((rng() % 100) < 32) w &= ~(1u << (rng() & 31)); }
        CHECK(hipMalloc(&d_bits, bits.size() * 4));
        CHECK(hipMemcpy(d_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
    }
    G1Group* d_groups;
    CHECK(hipMalloc(&d_pts, pts.size() * 4));
    CHECK(hipMalloc(&d_pts29, pts.size() * 4));
    CHECK(hipMalloc(&d_mem, members.size() * 4));
    CHECK(hipMemcpy(d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_mem, members.data(), members.size() * 4, hipMemcpyHostToDevice));
    launch_g1_table_s29(0, d_pts, d_pts29, NV);
    CHECK(hipDeviceSynchronize());
    const size_t lane_words = (size_t)(G1_LANE_PARTIAL_BYTES / 4) * 131072 * 2;
    CHECK(hipMalloc(&d_lane, lane_words * 4));
    CHECK(hipMalloc(&d_lane2, lane_words * 4));
    CHECK(hipMalloc(&d_wg, (size_t)G1X_WORDS * NG * 4 * 4));
    CHECK(hipMalloc(&d_groups, sizeof(G1Group) * NG));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (uint32_t k : {8u, 16u, 32u}) {
        const uint32_t tasks = (SIZE + k - 1) / k;
        uint32_t lb = 0;
        while ((1u << lb) < tasks) ++lb;
        if (lb > 8) continue;
        std::vector<G1Group> g(NG);
        for (uint32_t i = 0; i < NG; ++i) {
            g[i].member_start = i * SIZE; g[i].n_members = SIZE; g[i].bits_word = with_bits ? i * ((SIZE + 31) / 32) : NONE32; g[i].slot_base = i << lb;
            g[i].n_tasks = tasks; g[i].k = k; g[i].log2_block = lb; g[i].out_base = i;
        }
        const uint32_t n_slots = NG << lb;
        CHECK(hipMemcpy(d_groups, g.data(), sizeof(G1Group) * NG, hipMemcpyHostToDevice));
        for (int form = 1; form < 2; ++form) {
            std::vector<float> ms;
            for (int rep = 0; rep < 24; ++rep) {
                CHECK(hipEventRecord(e0));
                launch_g1_accumulate(0, d_pts29, d_mem, with_bits ? d_bits : nullptr, d_groups, NG, n_slots, d_lane2, d_wg, nullptr, nullptr);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float t; CHECK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            // and 24 launches in one go: the per-launch time with nothing between them
            CHECK(hipEventRecord(e0));
            for (int rep = 0; rep < 24; ++rep) {
                launch_g1_accumulate(0, d_pts29, d_mem, with_bits ? d_bits : nullptr, d_groups, NG, n_slots, d_lane2, d_wg, nullptr, nullptr);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float tot; CHECK(hipEventElapsedTime(&tot, e0, e1));
            std::sort(ms.begin(), ms.end());
            printf("%-6s k=%2u slots=%6u: first-by-one min %.1f med %.1f max %.1f us | back-to-back %.1f us/launch | %.2f G adds/s\n",
                   form ? "S29" : "12x32", k, n_slots, ms[0] * 1e3, ms[12] * 1e3, ms[23] * 1e3, tot / 24 * 1e3,
                   (double)(NV - n_slots) / (tot / 24) / 1e6);
        }
        {   // the tree over these partials, alone
            CHECK(hipEventRecord(e0));
            for (int rep = 0; rep < 24; ++rep) launch_g1_tree(0, d_lane2, d_groups, NG, n_slots, d_wg);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float tot; CHECK(hipEventElapsedTime(&tot, e0, e1));
            printf("       k_g1_tree over %u lane partials: %.1f us/launch\n", n_slots, tot / 24 * 1e3);
        }
    }
    return 0;
}
