// capi_shared.hpp -- what the host half (capi_host.cpp) and the device half (glome_device.hip) of the C ABI share.
#pragma once
#include <string>

#include "host_graph.hpp"

struct glome_sb {
  glome::Graph graph;
  std::string err;
};
inline const glome::Graph& sb_graph(const glome_sb* sb) { return sb->graph; }
