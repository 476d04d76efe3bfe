// Sanitizer harness for the host packer (bark_amd/csrc/pack.cpp): built by tests/test_packer_sanitizers.py with
// -fsanitize=address,undefined and with -fsanitize=thread.  Feeds valid forests, mutated forests and random bytes to
// bark_forest_pack_info / bark_forest_pack (B >= 32 spreads the forests over several host threads) and checks that
// the packer either succeeds within the buffer it announced or fails with a status and a message — never overruns,
// never races.  Prints "ok <n_valid> <n_rejected>".
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/bark_hip.h"

namespace {
constexpr int NODE = 26;
void put(uint8_t *rec, uint8_t leaf, uint32_t feat, float thr, uint32_t l, uint32_t r) {
    std::memset(rec, 0, NODE);
    rec[0] = leaf;
    std::memcpy(rec + 1, &feat, 4);
    std::memcpy(rec + 5, &thr, 4);
    std::memcpy(rec + 9, &l, 4);
    std::memcpy(rec + 13, &r, 4);
    rec[25] = 1;
}
// random binary tree grown in a container of L slots, children in arbitrary free slots
void grow(uint8_t *tree, int L, int d, std::mt19937 &rng, int max_nodes) {
    std::vector<int> free_slots;
    for (int i = 1; i < L; ++i) free_slots.push_back(i);
    std::shuffle(free_slots.begin(), free_slots.end(), rng);
    std::vector<int> leaves = {0};
    put(tree, 1, 0, 0.f, 0, 0);
    int nodes = 1;
    while (nodes + 2 <= max_nodes && free_slots.size() >= 2 && !leaves.empty()) {
        const size_t pick = rng() % leaves.size();
        const int n = leaves[pick];
        leaves.erase(leaves.begin() + (long)pick);
        const int l = free_slots.back();
        free_slots.pop_back();
        const int r = free_slots.back();
        free_slots.pop_back();
        const uint32_t f = rng() % (uint32_t)d;
        const float thr = f == 0 ? (float)(1 + rng() % 30) : (float)(rng() % 1000) / 1000.f;  // feature 0 is categorical
        put(tree + (size_t)n * NODE, 0, f, thr, (uint32_t)l, (uint32_t)r);
        put(tree + (size_t)l * NODE, 1, 0, 0.f, 0, 0);
        put(tree + (size_t)r * NODE, 1, 0, 0.f, 0, 0);
        leaves.push_back(l);
        leaves.push_back(r);
        nodes += 2;
    }
}
}  // namespace

int main() {
    std::mt19937 rng(12345);
    const int64_t d = 4;
    const int64_t ft[4] = {0, 1, 2, 2};
    long ok = 0, rejected = 0;
    for (int round = 0; round < 60; ++round) {
        const int64_t B = (round % 3 == 0) ? 48 : 1 + rng() % 5, m = 1 + rng() % 7, L = 4 + rng() % 60;
        std::vector<uint8_t> nodes((size_t)(B * m * L * NODE));
        for (auto &x : nodes) x = (uint8_t)rng();  // garbage in the unused slots
        for (int64_t t = 0; t < B * m; ++t) grow(nodes.data() + (size_t)t * L * NODE, (int)L, (int)d, rng, 1 + (int)(rng() % L));
        const int mode = round % 4;  // 0, 1: valid; 2: a few corrupted bytes; 3: random bytes everywhere
        if (mode == 2)
            for (int k = 0; k < 6; ++k) nodes[rng() % nodes.size()] = (uint8_t)rng();
        if (mode == 3)
            for (auto &x : nodes) x = (uint8_t)rng();
        bark_pack_info info;
        std::memset(&info, 0, sizeof info);
        int rc = bark_forest_pack_info(nodes.data(), B, m, L, ft, d, &info);
        if (rc != BARK_OK) {
            if (bark_last_error()[0] == 0) return 2;  // a failure must carry a message
            if (mode < 2) return 3;                   // valid forests must pack
            ++rejected;
            continue;
        }
        if (info.packed_bytes != B * m * info.stride * 16 || info.stride < 1 || info.stride > L || info.max_depth >= L) return 4;
        std::vector<uint8_t> packed((size_t)info.packed_bytes);  // exact size: ASan flags any overrun
        rc = bark_forest_pack(nodes.data(), ft, d, &info, packed.data());
        if (rc != BARK_OK) {
            if (mode < 2) return 5;
            ++rejected;
            continue;
        }
        // a mismatching info must be refused, not overrun the buffer
        bark_pack_info small = info;
        if (small.stride > 1) {
            small.stride -= 1;
            small.packed_bytes = B * m * small.stride * 16;
            std::vector<uint8_t> tight((size_t)small.packed_bytes);
            if (bark_forest_pack(nodes.data(), ft, d, &small, tight.data()) == BARK_OK && info.stride > small.stride) {
                // only acceptable if no tree actually needed the last slot
            }
        }
        ++ok;
    }
    std::printf("ok %ld %ld\n", ok, rejected);
    return 0;
}
