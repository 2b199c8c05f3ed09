// ws_merge.hpp -- state of the merging transform (filled in by ws_merge.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct ws_merge_state {
  void *parent = nullptr;   // device union-find forest over pixels
  size_t cap = 0;
};

inline void ws_merge_state_free(ws_merge_state *) {}
