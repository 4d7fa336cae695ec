// OutputBuffer.h — framebuffer owner with the protocol of the reference's
// sutil::CUDAOutputBuffer<PIXEL_FORMAT> (sutil/CUDAOutputBuffer.h:54-94): map() yields a device
// pointer valid until unmap(), getHostPointer() the pixels on the host.  Modes: DEVICE (device
// memory + explicit copy) and ZERO_COPY (pinned host memory mapped into the device); the two GL /
// P2P modes of the reference have no meaning on a headless node.
// Constructor as in the reference, (type, width, height): the buffer allocates in the *current* render
// context, the way CUDAOutputBuffer allocates on the current CUDA device (sutil/CUDAOutputBuffer.h:58,
// 100-137).  createDeviceContext() makes its context current; setContext() rebinds explicitly.
#pragma once
#include <cstdint>
#include <vector>
#include "Exception.h"

namespace acgpt {

enum class OutputBufferType { DEVICE = 0, ZERO_COPY = 2 };

// the render context OutputBuffer allocates in unless told otherwise (one per process, like the current device)
inline pt_ctx*& currentContext() { static pt_ctx* ctx = nullptr; return ctx; }
inline void makeContextCurrent(pt_ctx* ctx) { currentContext() = ctx; }

template <typename PIXEL_FORMAT>
class OutputBuffer {
public:
    OutputBuffer(OutputBufferType type, int32_t width, int32_t height) : m_ctx(currentContext()), m_type(type) { resize(width, height); }
    ~OutputBuffer() { release(); }
    OutputBuffer(const OutputBuffer&) = delete;
    OutputBuffer& operator=(const OutputBuffer&) = delete;

    void setDevice(int32_t) {}
    // move the buffer to another render context (re-allocates; contents are not carried over)
    void setContext(pt_ctx* ctx) { release(); m_ctx = ctx; resize(m_width, m_height); }
    void setStream(void* stream) { PT_CHECK(m_ctx, pt_set_stream(m_ctx, stream)); }

    void resize(int32_t width, int32_t height)
    {
        if (!m_ctx) throw Exception("OutputBuffer: no render context is current (createDeviceContext / makeContextCurrent first)");
        release();
        m_width = width < 1 ? 1 : width;
        m_height = height < 1 ? 1 : height;
        const size_t bytes = sizeof(PIXEL_FORMAT) * (size_t)m_width * m_height;
        if (m_type == OutputBufferType::DEVICE) {
            PT_CHECK(m_ctx, pt_device_malloc(m_ctx, &m_device_pixels, bytes));
        } else {
            PT_CHECK(m_ctx, pt_host_malloc_mapped(m_ctx, &m_host_zcopy_pixels, &m_device_pixels, bytes));
        }
        m_host_pixels.resize((size_t)m_width * m_height);
    }

    PIXEL_FORMAT* map() { return reinterpret_cast<PIXEL_FORMAT*>(m_device_pixels); }
    void unmap() {}   // pt_launch returns synchronised (the reference syncs its stream here)

    int32_t width() const { return m_width; }
    int32_t height() const { return m_height; }

    PIXEL_FORMAT* getHostPointer()
    {
        const size_t bytes = sizeof(PIXEL_FORMAT) * (size_t)m_width * m_height;
        if (m_type == OutputBufferType::DEVICE) {
            PT_CHECK(m_ctx, pt_copy_to_host(m_ctx, m_host_pixels.data(), m_device_pixels, bytes));
            return m_host_pixels.data();
        }
        return reinterpret_cast<PIXEL_FORMAT*>(m_host_zcopy_pixels);
    }

private:
    void release()
    {
        if (m_type == OutputBufferType::DEVICE) { if (m_device_pixels) pt_device_free(m_ctx, m_device_pixels); }
        else if (m_host_zcopy_pixels) pt_host_free_mapped(m_ctx, m_host_zcopy_pixels);
        m_device_pixels = nullptr; m_host_zcopy_pixels = nullptr;
    }
    pt_ctx* m_ctx;
    OutputBufferType m_type;
    int32_t m_width = 0, m_height = 0;
    void* m_device_pixels = nullptr;
    void* m_host_zcopy_pixels = nullptr;
    std::vector<PIXEL_FORMAT> m_host_pixels;
};

}  // namespace acgpt
