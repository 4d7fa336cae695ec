// ImageIO.h — PPM / PNG writers with the conventions of the reference's sutil::saveImage
// (sutil/sutil.cpp:542-655): the buffer's row 0 is the BOTTOM row, files are written top-down
// (vertical flip), UNSIGNED_BYTE4 pixels are taken as already sRGB-encoded, alpha is dropped.
#pragma once
#include <cstdint>
#include <string>

namespace acgpt {

// rgba: width*height*4 bytes, bottom-left origin.  Returns false on I/O failure or unknown suffix.
bool saveImage(const std::string& filename, const uint8_t* rgba, int width, int height);
bool savePPM(const std::string& filename, const uint8_t* rgba, int width, int height);
bool savePNG(const std::string& filename, const uint8_t* rgba, int width, int height);

}  // namespace acgpt
