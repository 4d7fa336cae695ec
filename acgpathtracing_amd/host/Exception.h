// Exception.h — error convention of the reference's sutil/Exception.h:82-112, 136-223 on top of
// the C ABI: a failing pt_* call becomes an acgpt::Exception carrying pt_last_error().
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
#include "../../include/acgpt.h"

namespace acgpt {

class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string& msg) : std::runtime_error(msg) {}
};

inline void ptCheck(int rc, pt_ctx* ctx, const char* call, const char* file, unsigned line)
{
    if (rc != 0) {
        std::stringstream ss;
        const char* m = pt_last_error(ctx);
        ss << "PT call '" << call << "' failed: " << (m ? m : "unknown") << " (" << file << ":" << line << ")";
        throw Exception(ss.str());
    }
}

}  // namespace acgpt

#define PT_CHECK(ctx, call) ::acgpt::ptCheck((call), (ctx), #call, __FILE__, __LINE__)
