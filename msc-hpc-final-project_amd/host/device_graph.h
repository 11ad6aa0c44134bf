// device_graph.h -- the device-resident form of an adjMatrix: one lzx handle per GPU the graph is spread over.
//
// The reference uploads IA/JA inside every cu_decompose() call (parallel-final/lib/cu_lanczos.cu:88-94) and, in its
// two-card variant, drives both cards from the one lanczosDecomp object (parallel-two-cards/lib/cu_lanczos.cu:39-191).
// Here the uploaded + reshaped graph belongs to the adjMatrix (built once: by the device ingest of the file
// constructor, or on the first device decomposition), and a lanczosDecomp borrows it.  Several GPUs: the handles are
// wired as an in-process communicator (lzx_comm_init_local), rows dealt by degree rank.
#pragma once

#include <vector>

struct lzx_ctx;

struct deviceGraph {
  std::vector<lzx_ctx *> ranks;   // rank r's handle; size() == number of GPUs (handles may share a GPU)
  double setup_ms = 0;            // upload / ingest + reshaping
  bool ingested = false;          // built by lzx_set_graph_edges (device ingest) rather than from host CSR arrays
  // The resident Lanczos basis belongs to one decomposition at a time: before another one overwrites it, the
  // previous owner is told to bring its basis to the host.
  void *owner = nullptr;
  void (*evict)(void *owner) = nullptr;
  deviceGraph() = default;
  deviceGraph(const deviceGraph &) = delete;
  deviceGraph &operator=(const deviceGraph &) = delete;
  ~deviceGraph();
};

// GPUs the drop-in classes place graphs on.  Default: environment LZX_DEVICES -- "all", a count ("4" = devices
// 0..3) or a list ("0,1,0": handles may share a GPU) -- else device 0.  An empty result = no usable GPU.
std::vector<int> lzx_host_devices();
void lzx_host_set_devices(const std::vector<int> &ids);   // overrides the environment (empty = back to it)
