/*
 * What the two translation units of the MP2 stage share (mp2_stage.hip: the kernels, the reference's decoder ABI, the batch;
 * mp2_live.hip: live streams, C ABI part 6).  Not installed; nothing outside jsmpeg_amd/csrc includes it.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "mp2_dev.h"

/* k_mp2_walk over n_streams streams; k_mp2_matrix / k_mp2_window over n_frames frames (frame places, for a live launch) */
hipError_t mp2_launch_walk(const Mp2Bufs &k, uint32_t n_streams, hipStream_t st);
hipError_t mp2_launch_matrix(const Mp2Bufs &k, uint32_t n_frames, hipStream_t st);
hipError_t mp2_launch_window(const Mp2Bufs &k, uint32_t n_frames, hipStream_t st);
/* the synthesis window D[0..511] on a device (uploaded once per device); 0 or < 0 */
int mp2_window_for_device(int dev, float **out);
