// tiles.hpp -- the reference's tile map (chunk / renderTiles, Glome.hs:371-386) and the wave decomposition of a tile.
#pragma once
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
// 64-pixel work items per tile: full 8x8 blocks, then the right / bottom leftovers packed 64 at a time
inline uint32_t tile_waves(int w, int h) {
  uint32_t nbx = w >> 3, nby = h >> 3;
  uint32_t rest = (uint32_t)(w * h) - nbx * nby * 64;
  return nbx * nby + (rest + 63) / 64;
}
// tiles owned by (first, stride) in renderTiles' order: x chunks outer, y chunks inner (Glome.hs:382-384)
inline void owned_tiles(int width, int height, int blocksize, int first, int stride, std::vector<DTile>& out, uint32_t& total_waves, int64_t& pixels) {
  out.clear(); total_waves = 0; pixels = 0;
  int k = 0;
  for (auto& xc : chunk(width, blocksize))
    for (auto& yc : chunk(height, blocksize)) {
      if (k >= first && (k - first) % stride == 0) {
        DTile t{xc.first, yc.first, xc.second, yc.second, total_waves, (uint32_t)pixels};
        out.push_back(t);
        total_waves += tile_waves(t.w, t.h);
        pixels += (int64_t)t.w * t.h;
      }
      k++;
    }
}

}  // namespace glome
