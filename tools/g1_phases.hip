// g1_phases.hip -- where does k_g1_accumulate spend its time?  Stamps wall_clock64() (100 MHz) in every workgroup at:
// start, end of wave 0's accumulation, after the first barrier (all waves accumulated), after each tree level.
// Workload = the bench's pubkey leg: 2048 groups x 512 members, k = 8 -> 64 tasks per group, all bits set
// (usage: g1_phases [k [members_per_group]]: 64 / 128 members = the per-GPU shard of configs[3] on 8 / 4 GPUs).
// Build (in tools/): hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPOSEVO_G1_PHASE_TIMING -I../pos_evolution_amd/csrc -o g1_phases g1_phases.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <random>
#include <vector>
#include "../pos_evolution_amd/csrc/g1_kernels.hip"

using namespace posevo;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const uint32_t n_groups = 2048, k = argc > 1 ? atoi(argv[1]) : 8, size = argc > 2 ? atoi(argv[2]) : 512;
    const uint32_t n_pts = n_groups * size;
    std::vector<uint32_t> pts((size_t)G1_ROW_WORDS * n_pts), members(n_pts);
    std::mt19937 rng(1);
    for (auto& w : pts) w = rng() & 0x0fffffffu;  // arbitrary field elements < p: the add formulas do not care
    for (uint32_t i = 0; i < n_pts; ++i) members[i] = i;
    std::shuffle(members.begin(), members.end(), rng);
    std::vector<G1Group> groups(n_groups);
    const uint32_t tasks = (size + k - 1) / k;
    uint32_t l2 = 0;
    while ((1u << l2) < tasks) ++l2;
    for (uint32_t g = 0; g < n_groups; ++g) {
        G1Group& d = groups[g];
        d.member_start = g * size; d.n_members = size; d.k = k; d.n_tasks = tasks; d.log2_block = l2;
        d.slot_base = g << l2; d.out_base = g; d.bits_word = NONE32;
    }
    const uint32_t n_slots = n_groups << l2;
    uint32_t *d_pts, *d_pts29, *d_members, *d_partials, *d_lane;
    G1Group* d_groups;
    CHECK(hipMalloc(&d_pts, pts.size() * 4));
    CHECK(hipMalloc(&d_members, members.size() * 4));
    CHECK(hipMalloc(&d_groups, sizeof(G1Group) * n_groups));
    CHECK(hipMalloc(&d_partials, 192ull * n_groups));
    CHECK(hipMalloc(&d_lane, 192ull * G1_WG * ((n_slots + G1_WG - 1) / G1_WG)));
    CHECK(hipMemcpy(d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_pts29, pts.size() * 4));
    launch_g1_table_s29(0, d_pts, d_pts29, n_pts);  // the accumulation reads the table in its own field form
    CHECK(hipMemcpy(d_members, members.data(), members.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_groups, groups.data(), sizeof(G1Group) * n_groups, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipEvent_t em;
    CHECK(hipEventCreate(&em));
    float ms = 0, ms_acc = 0;
    for (int rep = 0; rep < 5; ++rep) {  // the two kernels of a sum, back to back (k_g1_finish is not part of this tool)
        CHECK(hipEventRecord(e0, 0));
        launch_g1_accumulate(0, d_pts29, d_members, nullptr, d_groups, n_groups, n_slots, d_lane, d_partials);
        CHECK(hipEventRecord(em, 0));
        launch_g1_tree(0, d_lane, d_groups, n_groups, n_slots, d_partials);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipEventElapsedTime(&ms_acc, e0, em));
    }
    printf("k_g1_accumulate %.1f us + k_g1_tree %.1f us (events)\n", ms_acc * 1e3, (ms - ms_acc) * 1e3);
    const uint32_t wgs = (n_slots + G1_WG - 1) / G1_WG;
    std::vector<unsigned long long> st(16 * 4096);
    CHECK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g1_phase_stamps), st.size() * 8));
    const uint32_t nw = std::min<uint32_t>(wgs, 4096);
    unsigned long long t_min = ~0ull, t_max = 0;
    double sum[12] = {0};
    for (uint32_t w = 0; w < nw; ++w) {
        const unsigned long long* s = &st[16 * w];
        t_min = std::min(t_min, s[0]);
        t_max = std::max(t_max, s[10]);
        for (int i = 1; i <= 10; ++i) sum[i] += (double)(s[i] - s[i - 1]);
    }
    const double tick_us = 0.01;  // wall_clock64 = 100 MHz
    printf("k=%u tasks/group=%u slots=%u workgroups=%u kernel %.1f us (event)\n", k, tasks, n_slots, wgs, ms * 1e3);
    printf("first start -> last end: %.1f us\n", (t_max - t_min) * tick_us);
    const char* names[] = {"", "accumulate (wave 0)", "stage + wait for the other waves", "level 128", "level 64", "level 32",
                           "level 16", "level 8", "level 4", "level 2", "level 1"};
    for (int i = 1; i <= 10; ++i) printf("  %-34s avg %.2f us\n", names[i], sum[i] / nw * tick_us);
    // start skew: how many workgroups start late (second wave of workgroups)
    uint32_t late = 0;
    for (uint32_t w = 0; w < nw; ++w) if ((st[16 * w] - t_min) * tick_us > 20.0) ++late;
    printf("workgroups starting > 20 us after the first: %u of %u\n", late, nw);
    // distribution of the per-workgroup total and of its accumulate phase, overall and by blockIdx % 8 (XCD)
    std::vector<double> tot(nw), acc(nw), endt(nw);
    for (uint32_t w = 0; w < nw; ++w) {
        tot[w] = (st[16 * w + 10] - st[16 * w]) * tick_us;
        acc[w] = (st[16 * w + 2] - st[16 * w]) * tick_us;
        endt[w] = (st[16 * w + 10] - t_min) * tick_us;
    }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("total per workgroup: min %.1f p50 %.1f p90 %.1f max %.1f us;  accumulate+wait: min %.1f p50 %.1f p90 %.1f max %.1f us\n",
           pct(tot, 0), pct(tot, 0.5), pct(tot, 0.9), pct(tot, 1), pct(acc, 0), pct(acc, 0.5), pct(acc, 0.9), pct(acc, 1));
    for (int x = 0; x < 8; ++x) {
        std::vector<double> a, t, s0;
        for (uint32_t w = x; w < nw; w += 8) { a.push_back(acc[w]); t.push_back(endt[w]); s0.push_back((st[16 * w] - t_min) * tick_us); }
        printf("  blockIdx %% 8 == %d: start p50 %.1f max %.1f | accumulate+wait p50 %.1f max %.1f | end p50 %.1f max %.1f us\n", x,
               pct(s0, 0.5), pct(s0, 1), pct(a, 0.5), pct(a, 1), pct(t, 0.5), pct(t, 1));
    }
    // wave placement: HW_ID (gfx9 layout: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]) + XCC_ID
    std::vector<unsigned long long> wi(3 * 4 * 4096);
    CHECK(hipMemcpyFromSymbol(wi.data(), HIP_SYMBOL(g1_wave_info), wi.size() * 8));
    std::map<uint32_t, std::vector<double>> by_simd;   // key: xcc | se | sh | cu | simd
    std::map<uint32_t, int> wgs_per_cu;
    for (uint32_t w = 0; w < nw * 4; ++w) {
        const uint32_t hw = (uint32_t)wi[3 * w], xcc = (uint32_t)(wi[3 * w] >> 32) & 0xf;
        const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        by_simd[(cu_key << 2) | simd].push_back((wi[3 * w + 2] - wi[3 * w + 1]) * tick_us);
        if ((w & 3) == 0) wgs_per_cu[cu_key] += 1;
    }
    std::map<int, int> hist_cu, hist_simd;
    std::map<int, std::pair<double, int>> time_by_load;
    double first_sum = 0, second_sum = 0;
    int pairs = 0;
    for (auto& kv : wgs_per_cu) hist_cu[kv.second] += 1;
    for (auto& kv : by_simd) {
        hist_simd[(int)kv.second.size()] += 1;
        for (double t : kv.second) { time_by_load[(int)kv.second.size()].first += t; time_by_load[(int)kv.second.size()].second += 1; }
        if (kv.second.size() == 2) {
            first_sum += std::min(kv.second[0], kv.second[1]);
            second_sum += std::max(kv.second[0], kv.second[1]);
            ++pairs;
        }
    }
    printf("distinct CUs used: %zu;  workgroups per CU histogram:", wgs_per_cu.size());
    for (auto& kv : hist_cu) printf("  %d WG: %d CUs", kv.first, kv.second);
    printf("\nwaves per SIMD histogram:");
    for (auto& kv : hist_simd) printf("  %d waves: %d SIMDs", kv.first, kv.second);
    printf("\naccumulate time by waves on the SIMD:");
    for (auto& kv : time_by_load) printf("  %d waves: %.1f us", kv.first, kv.second.first / kv.second.second);
    if (pairs) printf("\nSIMDs with two waves: the earlier one finishes accumulating after %.1f us, the later after %.1f us (avg)",
                      first_sum / pairs, second_sum / pairs);
    printf("\n");
    return 0;
}
