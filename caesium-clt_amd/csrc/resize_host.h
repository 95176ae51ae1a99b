// resize_host.h -- host side of the Lanczos3 resize shared by the JPEG and the PNG batch objects (defined in pipeline.cpp)
#pragma once
#include <vector>

#include "types.h"

// image-rs Lanczos3 taps of one axis (imageops::sample): appends out_size taps and their normalised f32 weights
void csh_lanczos_axis(int in_size, int out_size, bool identity, std::vector<csh::ResizeTap> &taps, std::vector<float> &weights);
// libcaesium resize.rs compute_dimensions [UPSTREAM-RECALL]: both given -> exact; one given -> keep aspect, f32, round half away
void csh_compute_dimensions(int ow, int oh, int dw, int dh, int &nw, int &nh);
