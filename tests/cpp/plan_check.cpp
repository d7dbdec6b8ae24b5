// Host-only invariants of the work plan (csrc/ba_plan.cpp), compiled with g++ by
// tests/test_plan_invariants.py.  Scene: every landmark is seen by `win`
// consecutive poses in `n_cam` cameras (the C2..C4 structure), plus a few
// landmarks seen by many poses and a few never observed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "ba_plan.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      if (g_fail < 20) {                                  \
        std::printf("FAIL line %d: ", __LINE__);          \
        std::printf(__VA_ARGS__);                         \
        std::printf("\n");                                \
      }                                                   \
      ++g_fail;                                           \
    }                                                     \
  } while (0)

int main(int argc, char **argv) {
  const int n_pose = argc > 1 ? atoi(argv[1]) : 120, n_pt = argc > 2 ? atoi(argv[2]) : 30000;
  const int win = argc > 3 ? atoi(argv[3]) : 5, n_cam = argc > 4 ? atoi(argv[4]) : 2;
  const int drop_pct = argc > 5 ? atoi(argv[5]) : 0;  // observations dropped pseudo-randomly (masked groups)
  std::vector<uint8_t> pf(n_pose, 0), qf(n_pt, 0);
  for (int k = 0; k < 3; ++k) pf[k] = 1;
  for (int q = 0; q < n_pt; q += 97) qf[q] = 1;
  std::vector<int32_t> oc, op, oq;
  std::vector<double> uv;
  for (int q = 0; q < n_pt; ++q) {
    if (q % 1013 == 5) continue;  // never observed
    int j0 = (int)((long long)q * (n_pose - win + 1) / n_pt), w = win;
    if (q % 4001 == 7) { j0 = 0; w = n_pose; }  // seen by every pose
    for (int j = j0; j < j0 + w; ++j)
      for (int c = 0; c < n_cam; ++c) {
        const unsigned hsh = (unsigned)(q * 2654435761u) ^ (unsigned)(j * 40503u + c * 977u);
        if (drop_pct > 0 && !(j == j0 && c == 0) && (int)((hsh >> 7) % 100u) < drop_pct) continue;
        oc.push_back(c); op.push_back(j); oq.push_back(q); uv.push_back(q); uv.push_back(j); }
  }
  ba::PlanInput in;
  in.n_cam = n_cam; in.n_pose = n_pose; in.pose_fixed = pf.data(); in.n_pt = n_pt; in.pt_fixed = qf.data();
  in.n_obs = (int64_t)oc.size(); in.obs_cam = oc.data(); in.obs_pose = op.data(); in.obs_pt = oq.data(); in.obs_uv = uv.data();
  ba::Plan pl;
  const std::string err = ba::build_plan(in, pl);
  CHECK(err.empty(), "build_plan: %s", err.c_str());
  if (!err.empty()) return 1;

  // ---- landmark order is a permutation; optimised first, then fixed ----
  std::vector<int> seen(n_pt, 0);
  CHECK((int)pl.pt_user_of_int.size() == n_pt, "all points owned at world 1");
  for (int k = 0; k < (int)pl.pt_user_of_int.size(); ++k) {
    const int q = pl.pt_user_of_int[k];
    CHECK(q >= 0 && q < n_pt && !seen[q], "duplicate / bad point %d", q);
    seen[q] = 1;
    CHECK(pl.pt_int_of_user[q] == k, "inverse map broken at %d", k);
    CHECK((k < pl.M) == (qf[q] == 0), "optimised landmarks must come first (k=%d)", k);
  }
  // ---- observation list: landmark-major, every pair id in range, last writer only ----
  // (masked covisibility groups pad their members to the union pattern: padded slots
  //  carry uv = NaN; the real observations are all there, each exactly once)
  {
    int64_t real = 0;
    for (int64_t s2 = 0; s2 < pl.n_obs; ++s2) real += pl.obs_uv[2 * s2] == pl.obs_uv[2 * s2];
    CHECK(real == in.n_obs && pl.n_obs >= in.n_obs, "observation count: %lld real of %lld slots, %lld given",
          (long long)real, (long long)pl.n_obs, (long long)in.n_obs);
    int64_t pad = 0;
    for (uint8_t v : pl.pair_pad) pad += v;
    CHECK(pad == pl.n_pair_pad && (pl.pair_pad.empty() || (int64_t)pl.pair_pad.size() == pl.P), "padded pair flags");
  }
  for (int i = 0; i < pl.M; ++i) {
    std::set<int> poses;
    for (int64_t s = pl.lm_obs_ptr[i]; s < pl.lm_obs_ptr[i + 1]; ++s) {
      CHECK(pl.obs_idx[4 * s + 2] == i, "obs %lld not under its landmark", (long long)s);
      const int pid = pl.obs_idx[4 * s + 3];
      if (pid >= 0) {
        CHECK(pid >= pl.lm_pair_ptr[i] && pid < pl.lm_pair_ptr[i + 1], "pair id out of the landmark's range");
        CHECK(poses.insert(pl.pair_pose[pid]).second, "two writers for one pair");
        CHECK(pl.pair_pose[pid] == pl.obs_idx[4 * s + 1] && pl.pair_lm[pid] == i, "pair (pose, landmark) mismatch");
      }
    }
    CHECK((int64_t)poses.size() == pl.lm_pair_ptr[i + 1] - pl.lm_pair_ptr[i], "every pair has exactly one writer");
  }
  // ---- Schur super-runs ----
  CHECK(pl.sup_lane.size() == pl.sup_desc.size() * 256, "lane table size");
  std::vector<int> covered(pl.M, 0);           // 1: all of the landmark's triples are in runs / groups
  std::vector<int64_t> run_triples(pl.M, 0);   // triples of the landmark filed in super-run chunks
  std::set<std::pair<int64_t, int64_t>> filed; // (pair p, pair q): every triple at most once
  int64_t triples = 0;
  for (size_t r = 0; r < pl.sup_desc.size(); ++r) {
    const auto &sd = pl.sup_desc[r];
    CHECK(sd.ns >= 1 && sd.ns <= ba::kSchurSlots, "slots per run");
    CHECK(sd.chunk_end - sd.chunk_begin >= 1 && sd.chunk_end - sd.chunk_begin <= ba::kSchurSuperChunks, "chunks per run");
    // lane table: every slot owns an even number (2..32) of consecutive lanes of ONE wave
    std::vector<int> first(sd.ns, -1), count(sd.ns, 0);
    for (int lane = 0; lane < 256; ++lane) {
      const uint32_t w = pl.sup_lane[r * 256 + lane];
      const int slot = w & 0xff;
      if (slot == 0xff) continue;
      CHECK(slot < sd.ns, "lane word names slot %d of %d", slot, sd.ns);
      if (slot >= sd.ns) continue;
      const int h = (w >> 8) & 1, sub2 = (w >> 9) & 0x1f, tps2 = (w >> 14) & 0x3f;
      if (first[slot] < 0) first[slot] = lane;
      const int sub = lane - first[slot];
      CHECK(sub == 2 * sub2 + h, "lane layout inside a slot (run %zu lane %d)", r, lane);
      CHECK(sub2 < tps2 && tps2 >= 1 && tps2 <= 16, "sub2 / tps2 range");
      CHECK(first[slot] / 64 == lane / 64, "slot straddles two waves");
      count[slot]++;
    }
    for (int s = 0; s < sd.ns; ++s) {
      CHECK(count[s] >= 2 && count[s] % 2 == 0 && count[s] <= 32, "slot %d has %d lanes", s, count[s]);
      const uint32_t w = pl.sup_lane[r * 256 + (first[s] < 0 ? 0 : first[s])];
      CHECK((int)((w >> 14) & 0x3f) * 2 == count[s], "tps2 field does not match the lane count");
    }
    for (int c = sd.chunk_begin; c < sd.chunk_end; ++c) {
      const auto &cd = pl.chunk_desc[c];
      CHECK(cd.np >= 1 && cd.np <= ba::kSchurPairs && cd.nl >= 1 && cd.nl <= ba::kSchurLandmarks &&
                cd.nt >= 1 && cd.nt <= ba::kSchurTri, "chunk limits");
      CHECK((cd.tb & 3) == 0 && cd.tb + cd.nt + 3 < (int64_t)pl.ltri.size(), "16-byte triple loads stay inside");
      CHECK(cd.p0 == pl.lm_pair_ptr[cd.l0], "chunk pair base");
      const uint16_t *sp = &pl.chunk_sp[cd.sp];
      CHECK(sp[0] == 0 && sp[sd.ns] == cd.nt, "slot offsets span the chunk's triples");
      for (int s = 0; s < sd.ns; ++s) {
        CHECK(sp[s] <= sp[s + 1], "slot offsets ascend");
        for (int t = sp[s]; t < sp[s + 1]; ++t) {
          const uint32_t w = pl.ltri[cd.tb + t];
          const int p = w >> 16, q = w & 0xff, lm = (w >> 8) & 0xff;
          CHECK(p < cd.np && q < cd.np && p <= q && lm < cd.nl, "triple word fields");
          CHECK(pl.pair_lm[cd.p0 + p] == cd.l0 + lm && pl.pair_lm[cd.p0 + q] == cd.l0 + lm, "triple's landmark");
          CHECK(filed.insert({cd.p0 + p, cd.p0 + q}).second, "triple filed twice");
          run_triples[cd.l0 + lm]++;
          const int jb = pl.pair_pose[cd.p0 + p], kb = pl.pair_pose[cd.p0 + q];
          const int blk = pl.slot_blk[sd.s0 + s];
          CHECK(pl.sblk_j[blk] == std::min(jb, kb) && pl.sblk_k[blk] == std::max(jb, kb), "triple filed under the wrong block");
        }
      }
      triples += cd.nt;
    }
  }
  // a landmark is in the runs with ALL its triples (whole, or split into pose-group
  // classes over several runs) or with none
  for (int l = 0; l < pl.M; ++l) {
    const int64_t d = pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l];
    CHECK(run_triples[l] == 0 || run_triples[l] == d * (d + 1) / 2, "landmark %d: %lld of %lld triples in runs", l,
          (long long)run_triples[l], (long long)(d * (d + 1) / 2));
    covered[l] = run_triples[l] > 0;
  }
  // ---- covisibility groups: the first M_grp landmarks, group after group ----
  int64_t grp_triples = 0;
  {
    int next = 0;
    for (const auto &gr : pl.grp_range) {
      CHECK(gr.l0 == next && gr.nl >= ba::kGrpMinLandmarks && gr.d >= 0 && gr.d <= ba::kGrpMaxPoses, "group range");
      next = gr.l0 + gr.nl;
    }
    // (groups of landmarks seen by fixed poses only, d = 0, have no Schur piece)
    std::vector<int> want_piece(pl.M, 0);
    for (const auto &gr : pl.grp_range)
      for (int l = gr.l0; l < gr.l0 + gr.nl; ++l) want_piece[l] = gr.d > 0;
    CHECK(next == pl.M_grp && pl.M_grp <= pl.M, "groups tile [0, M_grp)");
    std::vector<int> gcov(pl.M, 0);
    for (const auto *list : {&pl.grp32, &pl.grp64, &pl.grp128})
      for (const auto &gd : *list) {
        CHECK((gd.d <= 5) == (list == &pl.grp32) && (gd.d > 10) == (list == &pl.grp128), "group filed under the wrong tile width");
        CHECK(gd.nl >= 1 && gd.nl <= ba::kGrpMaxLandmarks && gd.l0 >= 0 && gd.l0 + gd.nl <= pl.M_grp, "group piece range");
        CHECK(gd.p0 == pl.lm_pair_ptr[gd.l0], "group pair base");
        for (int t = 1; t < gd.d; ++t) CHECK(gd.pose[t - 1] < gd.pose[t], "group poses ascend");
        for (int l = gd.l0; l < gd.l0 + gd.nl; ++l) {
          gcov[l]++;
          CHECK(pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l] == gd.d, "group landmark degree");
          for (int t = 0; t < gd.d; ++t) {
            const int64_t p = gd.p0 + (int64_t)gd.d * (l - gd.l0) + t;
            CHECK(pl.pair_lm[p] == l && pl.pair_pose[p] == gd.pose[t], "pair(il, jj) = p0 + d il + jj");
          }
        }
        int idx = 0;
        for (int jj = 0; jj < gd.d; ++jj)
          for (int kk = jj; kk < gd.d; ++kk, ++idx) {
            CHECK(idx == jj * gd.d - jj * (jj - 1) / 2 + (kk - jj), "slot index formula of k_schur_grp");
            const int blk = pl.slot_blk[gd.s0 + idx];
            CHECK(pl.sblk_j[blk] == gd.pose[jj] && pl.sblk_k[blk] == gd.pose[kk], "group slot block");
          }
        grp_triples += (int64_t)gd.nl * gd.d * (gd.d + 1) / 2;
      }
    // k_lin_grp pieces: tile the grouped landmarks; slot oo of every landmark of a piece
    // is an observation by pattern pose / camera oo, the slots of a pose are adjacent, its
    // last slot writes the pair
    {
      int nextl = 0;
      long long n_masked_slots = 0;
      // (plain pieces first, masked pieces behind them: the tiling is by landmark)
      std::vector<ba::Plan::LinDesc> pieces(pl.lin_desc.begin(), pl.lin_desc.end());
      for (size_t k = 0; k < pieces.size(); ++k)
        CHECK((pieces[k].pad_ == 0) == ((int)k < pl.n_lin_plain), "plain pieces precede the masked ones");
      std::sort(pieces.begin(), pieces.end(), [](const ba::Plan::LinDesc &x, const ba::Plan::LinDesc &y) { return x.l0 < y.l0; });
      for (const auto &gd : pieces) {
        CHECK(gd.l0 == nextl && gd.nl >= 1, "k_lin_grp pieces tile [0, M_grp)");
        nextl = gd.l0 + gd.nl;
        CHECK(gd.no >= gd.d && gd.no >= 1 && gd.no <= ba::kGrpMaxObs && gd.o0 == pl.lm_obs_ptr[gd.l0] && gd.p0 == pl.lm_pair_ptr[gd.l0],
              "piece observation / pair base");
        CHECK(gd.nl <= ba::kLinGrpSteps * 4 * ba::lin_grp_nlw(gd.no) || getenv("BA_LIN_STEPS"), "piece size");
        CHECK(gd.pat0 >= 0 && (size_t)(gd.pat0 + gd.no) * 2 <= pl.grp_pat.size(), "group pattern range");
        int pose_valid[ba::kGrpMaxPoses] = {0};
        for (int l = gd.l0; l < gd.l0 + gd.nl; ++l) {
          CHECK(pl.lm_obs_ptr[l + 1] - pl.lm_obs_ptr[l] == gd.no, "group landmark observation count");
          CHECK(pl.lm_pair_ptr[l + 1] - pl.lm_pair_ptr[l] == gd.d, "group landmark degree");
          for (int oo = 0; oo < gd.no; ++oo) {
            const int64_t s = gd.o0 + (int64_t)gd.no * (l - gd.l0) + oo;
            const int32_t pj = pl.grp_pat[2 * (gd.pat0 + oo)], w = pl.grp_pat[2 * (gd.pat0 + oo) + 1];
            const int cam = w & 0xffff, jj = (w >> 16) & 0xff, opt = (w >> 29) & 1, lastw = (w >> 30) & 1;
            CHECK(pl.obs_idx[4 * s + 0] == cam && pl.obs_idx[4 * s + 1] == pj && pl.obs_idx[4 * s + 2] == l,
                  "obs(il, oo) = o0 + no il + oo follows the pattern");
            CHECK(opt == (pj < pl.N), "pattern optimisable flag");
            if (opt) CHECK(jj < gd.d && pl.pair_pose[gd.p0 + jj] == pj, "pattern pose slot");
            const int64_t pair = pl.obs_idx[4 * s + 3];
            CHECK(lastw == (pair >= 0), "pattern last-writer flag");
            if (lastw) CHECK(pair == gd.p0 + (int64_t)gd.d * (l - gd.l0) + jj, "pattern pair id");
            const bool have = pl.obs_uv[2 * s] == pl.obs_uv[2 * s];
            if (!gd.pad_) CHECK(have, "a plain group has no padded slot");
            if (opt && have) pose_valid[jj] = 1;
            n_masked_slots += !have;
          }
          for (int jj = 0; jj < gd.d; ++jj) {
            const int64_t pr = gd.p0 + (int64_t)gd.d * (l - gd.l0) + jj;
            const bool pad = !pl.pair_pad.empty() && pl.pair_pad[pr];
            CHECK(pad == !pose_valid[jj], "a pair is padded iff its pose has no valid slot in the landmark");
            pose_valid[jj] = 0;
          }
        }
      }
      CHECK(nextl == (pl.lin_groups ? pl.M_grp : 0), "k_lin_grp pieces cover the grouped landmarks");
      std::printf("masked groups: %d pieces of %zu, %lld padded slots, %lld padded pairs\n",
                  (int)pl.lin_desc.size() - pl.n_lin_plain, pl.lin_desc.size(), n_masked_slots, (long long)pl.n_pair_pad);
      if (drop_pct > 0 && !getenv("BA_NO_SUPERSET") && pl.lin_groups)
        CHECK((int)pl.lin_desc.size() > pl.n_lin_plain && n_masked_slots > 0, "the dropout scene must produce masked groups");
    }
    for (int l = 0; l < pl.M; ++l) {
      CHECK(gcov[l] == want_piece[l], "landmark %d in %d group pieces", l, gcov[l]);
      CHECK(!(gcov[l] && covered[l]), "landmark %d in a group and in a super-run", l);
      covered[l] += gcov[l];
    }
    // every slot is named by exactly one contribution entry of its block
    std::vector<int> slot_seen(pl.slot_blk.size(), 0);
    for (int64_t bk = 0; bk < pl.B; ++bk)
      for (int64_t c = pl.blk_contrib_ptr[bk]; c < pl.blk_contrib_ptr[bk + 1]; ++c) {
        CHECK(pl.slot_blk[pl.contrib_slot[c]] == bk, "contribution list names a foreign slot");
        slot_seen[pl.contrib_slot[c]]++;
      }
    for (size_t sl = 0; sl < slot_seen.size(); ++sl) CHECK(slot_seen[sl] == 1, "slot %zu summed %d times", sl, slot_seen[sl]);
  }
  triples += grp_triples;
  // every landmark with pairs is covered exactly once, by a group, a super-run or the big-landmark list
  int64_t big_triples = 0;
  for (int i = 0; i < pl.M; ++i) {
    const int64_t d = pl.lm_pair_ptr[i + 1] - pl.lm_pair_ptr[i];
    if (d == 0) continue;
    const bool big = covered[i] == 0;
    CHECK(covered[i] <= 1, "landmark %d in %d chunks", i, covered[i]);
    if (big) big_triples += d * (d + 1) / 2;
  }
  CHECK(triples + big_triples == pl.T, "triples: %lld in runs + %lld big != %lld", (long long)triples,
        (long long)big_triples, (long long)pl.T);
  CHECK((int64_t)pl.tri_p.size() == big_triples, "global triple list holds exactly the big landmarks");
  // back-substitution chunks tile the landmarks
  CHECK(pl.bchunk_lm.front() == 0 && pl.bchunk_lm.back() == pl.M, "backsub chunks cover [0, M)");
  CHECK(pl.n_bchunk_grp >= 0 && pl.n_bchunk_grp < (int)pl.bchunk_lm.size() && pl.bchunk_lm[pl.n_bchunk_grp] == pl.M_grp,
        "no chunk straddles M_grp");
  // the pose-major list holds exactly the observations k_lin_grp does not take
  {
    int64_t want = 0;
    for (int64_t s = 0; s < pl.n_obs; ++s)
      if (pl.obs_idx[4 * s + 1] < pl.N && !(pl.lin_groups && pl.obs_idx[4 * s + 2] < pl.M_grp)) ++want;
    CHECK(want == pl.n_pobs, "pose-major list size %lld != %lld", (long long)pl.n_pobs, (long long)want);
    // rows of Apart2: every (piece, pose) exactly once in its pose's list
    std::vector<int> seen((size_t)pl.n_apart2, 0);
    for (int j = 0; j < pl.N; ++j)
      for (int q = pl.pose_gpart_ptr[j]; q < pl.pose_gpart_ptr[j + 1]; ++q) seen[pl.pose_gpart[q]]++;
    for (size_t r = 0; r < seen.size(); ++r) CHECK(seen[r] == 1, "Apart2 row %zu listed %d times", r, seen[r]);
    for (const auto &gd : pl.lin_desc)
      for (int t = 0; t < gd.d; ++t) {
        const int pj = pl.pair_pose[gd.p0 + t];
        bool found = false;
        for (int q = pl.pose_gpart_ptr[pj]; q < pl.pose_gpart_ptr[pj + 1]; ++q) found |= pl.pose_gpart[q] == gd.apart0 + t;
        CHECK(found, "Apart2 row of (piece, pose) missing from the pose's list");
      }
    std::vector<int> cseen(pl.bchunk_lm.size() - 1 + pl.lin_desc.size(), 0);
    for (const auto &gd : pl.lin_desc) cseen[gd.cost_idx]++;
    for (size_t c = pl.bchunk_lm.size() - 1; c < cseen.size(); ++c) CHECK(cseen[c] == 1, "cost partial entry %zu", c);
  }
  for (size_t c = 0; c + 1 < pl.bchunk_lm.size(); ++c)
    CHECK(pl.bchunk_lm[c] < pl.bchunk_lm[c + 1] && pl.bchunk_lm[c + 1] - pl.bchunk_lm[c] <= ba::kSchurLandmarks,
          "backsub chunk size");
  // the cost kernel's slim record mirrors the landmark-major list
  CHECK(pl.obs_cp.size() == (size_t)pl.n_obs * 2, "slim record list size");
  for (int64_t k = 0; k < pl.n_obs; ++k)
    CHECK(pl.obs_cp[2 * k] == (pl.obs_idx[4 * k] | (pl.obs_idx[4 * k + 1] << 16)) && pl.obs_cp[2 * k + 1] == pl.obs_idx[4 * k + 2],
          "slim record differs from the landmark-major record");
  // ---- checksum of every array the kernels consume: the test runs this program with
  // BA_PLAN_THREADS = 1 and = 7 and requires the same plan ----
  {
    auto mix = [](uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ull + (h >> 29); };
    uint64_t h = 1469598103934665603ull;
    for (auto v : pl.obs_idx) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.obs_cp) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.obs_uv) { uint64_t b; memcpy(&b, &v, 8); h = mix(h, b); }
    for (auto v : pl.pair_pose) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.pair_lm) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.lm_obs_ptr) h = mix(h, (uint64_t)v);
    for (auto v : pl.lm_pair_ptr) h = mix(h, (uint64_t)v);
    for (auto v : pl.pt_user_of_int) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.pobs_idx) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.sblk_j) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.sblk_k) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.ltri) h = mix(h, (uint64_t)v);
    for (auto v : pl.grp_pat) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto v : pl.contrib_slot) h = mix(h, (uint64_t)(uint32_t)v);
    for (auto &g : pl.lin_desc) h = mix(mix(mix(h, (uint64_t)g.p0), (uint64_t)g.o0), (uint64_t)(uint32_t)g.l0);
    std::printf("plan checksum %016llx\n", (unsigned long long)h);
  }
  std::printf("plan: M=%d (grouped %d in %zu+%zu pieces) P=%lld runs=%zu chunks=%zu triples=%lld (%lld in groups, +%lld big)  %s\n",
              pl.M, pl.M_grp, pl.grp32.size(), pl.grp64.size() + pl.grp128.size(), (long long)pl.P, pl.sup_desc.size(),
              pl.chunk_desc.size(), (long long)triples, (long long)grp_triples, (long long)big_triples,
              g_fail ? "FAILED" : "OK");
  return g_fail ? 1 : 0;
}
