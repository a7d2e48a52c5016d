// image.hpp — the host-side page table behind the opaque bx_image of include/bx_image.h (shared by image_host.cpp and image.hip).
#pragma once
#include <stdint.h>

#include <map>
#include <vector>

#include "../../include/bx_image.h"

struct bx_image {
    std::map<uint32_t, std::vector<uint32_t>> pages;  // page index -> 256 words; an absent page is all zeros
};
