// tiles.hpp -- the reference's tile map (chunk / renderTiles, Glome.hs:371-386) and the wave decomposition of a tile.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "rt_types.h"

namespace glome {

inline std::vector<std::pair<int, int>> chunk(int size, int blocksize) {  // Glome.hs:371-377: the last chunk is the remainder
  std::vector<std::pair<int, int>> o;
  int pos = 0;
  for (;;) {
    if (pos + blocksize >= size) { o.push_back({pos, size - pos}); break; }
    o.push_back({pos, blocksize});
    pos += blocksize;
  }
  return o;
}
// 64-pixel work items per tile: full kBlockW x kBlockH blocks (8x8), then the right / bottom leftovers packed 64 at a time
inline uint32_t tile_waves(int w, int h) {
  uint32_t nbx = w / kBlockW, nby = h / kBlockH;
  uint32_t rest = (uint32_t)(w * h) - nbx * nby * 64;
  return nbx * nby + (rest + 63) / 64;
}
// Who owns tile k of a frame shared by `world` ranks.  Plain round robin (share_pct 0 or 100): rank k mod world.  With
// rank0_share_pct in 1..99 rank 0 -- the rank that also receives and blits every frame -- gets that WEIGHT, in percent of
// ONE other rank's: rank 0 owns pct / (pct + 100 * (world - 1)) of the tiles, every other rank 100 / (pct + 100 * (world - 1)).
// The pattern repeats with period (pct + 100 * (world - 1)) / gcd(pct, 100) slots (pct 80, two ranks: 4 + 5 = 9), each rank's
// slots spread evenly over the period (largest remaining deficit first), so neighbouring tiles still go to different ranks.
// Every rank computes the same pattern from (world, pct); every percent is a different layout.
inline std::vector<int> shard_pattern(int world, int share_pct) {
  std::vector<int> pat;
  if (world <= 1 || share_pct <= 0 || share_pct >= 100) { for (int r = 0; r < std::max(world, 1); r++) pat.push_back(r); return pat; }
  int g = share_pct, h = 100;
  while (h) { const int t = g % h; g = h; h = t; }
  const int s = 100 / g, s0 = share_pct / g, period = s0 + (world - 1) * s;
  std::vector<long> given((size_t)world, 0);
  for (int j = 0; j < period; j++) {
    int best = 0; long bestd = -(1L << 60);
    for (int r = 0; r < world; r++) {
      const long target = r == 0 ? s0 : s;
      const long d = target * (j + 1) - given[(size_t)r] * period;  // how far behind its share rank r is after slot j
      if (d > bestd) { bestd = d; best = r; }
    }
    given[(size_t)best]++;
    pat.push_back(best);
  }
  return pat;
}
// tiles owned by (first, stride) in renderTiles' order: x chunks outer, y chunks inner (Glome.hs:382-384)
inline void owned_tiles(int width, int height, int blocksize, int first, int stride, int share_pct, std::vector<DTile>& out, uint32_t& total_waves, int64_t& pixels) {
  out.clear(); total_waves = 0; pixels = 0;
  const bool weighted = share_pct > 0 && share_pct < 100 && stride > 1 && first < stride;
  std::vector<int> pat;
  if (weighted) pat = shard_pattern(stride, share_pct);
  int k = 0;
  for (auto& xc : chunk(width, blocksize))
    for (auto& yc : chunk(height, blocksize)) {
      const bool mine = weighted ? pat[(size_t)k % pat.size()] == first : (k >= first && (k - first) % stride == 0);
      if (mine) {
        DTile t{xc.first, yc.first, xc.second, yc.second, total_waves, (uint32_t)pixels};
        out.push_back(t);
        total_waves += tile_waves(t.w, t.h);
        pixels += (int64_t)t.w * t.h;
      }
      k++;
    }
}

}  // namespace glome
