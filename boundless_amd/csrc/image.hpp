// image.hpp — the host-side page table behind the opaque bx_image of include/bx_image.h (shared by image_host.cpp and image.hip).
#pragma once
#include <stdint.h>

#include <map>
#include <vector>

#include "../../include/bx_image.h"

#include <array>

struct bx_image {
    std::map<uint32_t, std::vector<uint32_t>> pages;  // page index -> 256 words; an absent page is all zeros
    // A PARTIAL image (the `partial_image` of a Segment: the pages a segment touches plus the digests of the subtrees it does
    // not): node index -> digest (8 canonical words).  Nodes are numbered as in risc0-binfmt's MemoryImage: root 1, children of i
    // are 2i and 2i+1, page p is node 2^22 + p.  A subtree given by its digest holds no pages and no further digests.
    std::map<uint32_t, std::array<uint32_t, 8>> digests;
};
