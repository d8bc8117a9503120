// image.hpp -- build the flat device image (image_layout.h) from a parsed `.lph`.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "image_layout.h"
#include "lph_format.hpp"

namespace lphq {
bool build_image(const LphFile& file, std::vector<uint8_t>& image, std::string& err);
}
