"""Host-side mirror of the reference's render path, over the C ABI.

Names, argument meaning and error behaviour follow PathTracer_Optix/PathTracerMain.cpp so a
test written against the reference's functions reads the same here:

    reference (PathTracerMain.cpp)                 here
    ------------------------------------------------------------------------------
    TinyObjWrapper obj(path)            :650       TinyObjWrapper(path)
    initCamera()                        :228-233   initCamera()
    createDeviceContext(state)          :240-258   createDeviceContext(state)
    buildTheAccelarationStructure(..)   :260-398   buildTheAccelarationStructure(state, obj)
    createModule/ProgramGroups/Pipeline :400-539   (nothing to do: AOT gfx950 code object)
    createShaderBindingTable(state,obj) :544-627   createShaderBindingTable(state, obj)
    initializeTheLaunch(state)          :143-164   initializeTheLaunch(state)
    updateState(output_buffer, state)   :166-182   updateState(output_buffer, state)
    LaunchCurrentFrame(buffer, state)   :184-210   LaunchCurrentFrame(output_buffer, state)
    keyCallback(...)                    :100-141   keyCallback(state, key)
    CleanAllTheThings(state)            :629-646   CleanAllTheThings(state)
    sutil::CUDAOutputBuffer<uchar4>                OutputBuffer (DEVICE / ZERO_COPY modes)

Everything that computes runs in libacgpt_hip.so; this module only marshals.
"""
import ctypes as C
import os

import numpy as np

from . import _native
from ._native import AreaLight, BvhInfo, Float3, Material, PathTraceParams, Stats

BSDF_DIFFUSE, BSDF_METALLIC, BSDF_REFRACTION = 0, 1, 2

# PathTracerMain.cpp:42-43, 58-59
maxiumumRecursionDepth = 28
samples_per_launch = 128

SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")


class PathTracerError(RuntimeError):
    """Counterpart of sutil::Exception thrown by CUDA_CHECK / OPTIX_CHECK (sutil/Exception.h:82-112)."""


def _check(ctx, rc, what):
    if rc != 0:
        msg = _native.hip().pt_last_error(ctx)
        raise PathTracerError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))


def _f3(v):
    return Float3(float(v[0]), float(v[1]), float(v[2]))


# ------------------------------------------------------------------ scene ingest ----
class TinyObjWrapper:
    """OBJ/MTL ingest, API of PathTracer_Optix/TinyObjWrapper.h:77-115 (the parsing itself is
    the C++ in acgpathtracing_amd/host/TinyObjWrapper.cpp)."""

    def __init__(self, filename=None):
        self.dataLoaded = False
        self._vertices = np.zeros(0, np.float32)
        self._indexBuffer = np.zeros(0, np.uint32)
        self._materialIndices = np.zeros(0, np.uint32)
        self._materials = (Material * 0)()
        self.warning = ""
        self.error = ""
        if filename is not None:
            self.loadFile(filename)

    def loadFile(self, filename):
        L = _native.host()
        h = L.pth_obj_load(os.fsencode(filename))
        try:
            self.dataLoaded = bool(L.pth_obj_ok(h))
            self.warning = (L.pth_obj_warning(h) or b"").decode()
            self.error = (L.pth_obj_error(h) or b"").decode()
            if not self.dataLoaded:
                return False
            sizes = [C.c_size_t() for _ in range(4)]
            L.pth_obj_sizes(h, *[C.byref(s) for s in sizes])
            nv, ni, nm, nmat = [s.value for s in sizes]
            self._vertices = np.zeros(nv, np.float32)
            self._indexBuffer = np.zeros(ni, np.uint32)
            self._materialIndices = np.zeros(nm, np.uint32)
            self._materials = (Material * nmat)()
            L.pth_obj_fill(h, self._vertices.ctypes.data, self._indexBuffer.ctypes.data,
                           self._materialIndices.ctypes.data, C.addressof(self._materials) if nmat else None)
        finally:
            L.pth_obj_free(h)
        return True

    def getVerticesFloat(self):
        return self._vertices

    def getIndexBuffer(self):
        return self._indexBuffer

    def getMaterialIndices(self):
        return self._materialIndices

    def getMaterials(self):
        return self._materials

    def getNumMaterials(self):
        return len(self._materials)


# ------------------------------------------------------------------------ camera ----
class Camera:
    """sutil::Camera (sutil/Camera.h:38-75); UVWFrame is computed by the C++ host library."""

    def __init__(self, eye=(1.0, 1.0, 1.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovY=35.0, aspectRatio=1.0):
        self.m_eye, self.m_lookat, self.m_up = tuple(eye), tuple(lookat), tuple(up)
        self.m_fovY, self.m_aspectRatio = float(fovY), float(aspectRatio)

    def eye(self): return self.m_eye
    def setEye(self, v): self.m_eye = tuple(v)
    def lookat(self): return self.m_lookat
    def setLookat(self, v): self.m_lookat = tuple(v)
    def up(self): return self.m_up
    def setUp(self, v): self.m_up = tuple(v)
    def fovY(self): return self.m_fovY
    def setFovY(self, v): self.m_fovY = float(v)
    def aspectRatio(self): return self.m_aspectRatio
    def setAspectRatio(self, v): self.m_aspectRatio = float(np.float32(v))

    def UVWFrame(self):
        e = np.asarray(self.m_eye, np.float32); l = np.asarray(self.m_lookat, np.float32); u = np.asarray(self.m_up, np.float32)
        U = np.zeros(3, np.float32); V = np.zeros(3, np.float32); W = np.zeros(3, np.float32)
        _native.host().pth_camera_uvw(e.ctypes.data, l.ctypes.data, u.ctypes.data, C.c_float(self.m_fovY),
                                      C.c_float(self.m_aspectRatio), U.ctypes.data, V.ctypes.data, W.ctypes.data)
        return U, V, W


g_camera = Camera()


def initCamera():
    """PathTracerMain.cpp:228-233."""
    g_camera.setEye((278.0, 273.0, -900.0))
    g_camera.setLookat((278.0, 273.0, 330.0))
    g_camera.setUp((0.0, 1.0, 0.0))
    g_camera.setFovY(35.0)
    return g_camera


# ----------------------------------------------------------------- output buffer ----
class OutputBufferType:
    """sutil::CUDAOutputBufferType (sutil/CUDAOutputBuffer.h:45-51); the GL_INTEROP and CUDA_P2P
    modes have no meaning on a headless single-process-per-GPU node."""
    DEVICE = 0
    ZERO_COPY = 2


class OutputBuffer:
    """uchar4 framebuffer owner with sutil::CUDAOutputBuffer's map/unmap/getHostPointer protocol."""

    def __init__(self, buffer_type, width, height, state=None):
        self.m_type = buffer_type
        self.m_width = self.m_height = 0
        self._state = state
        self._dev = None
        self._host_mapped = None
        self._host = None
        if state is not None:
            self.resize(width, height)
        else:
            self.m_width, self.m_height = int(width), int(height)

    def _ctx(self):
        if self._state is None or not self._state.context:
            raise PathTracerError("OutputBuffer: no device context")
        return self._state.context

    def attach(self, state):
        self._state = state
        self.resize(self.m_width, self.m_height)

    def width(self): return self.m_width
    def height(self): return self.m_height
    def setStream(self, stream): pass     # launches and copies share the context's stream
    def setDevice(self, device_idx): pass

    def _release(self):
        L = _native.hip()
        if self._dev is not None and self.m_type == OutputBufferType.DEVICE:
            L.pt_device_free(self._ctx(), self._dev)
        if self._host_mapped is not None:
            L.pt_host_free_mapped(self._ctx(), self._host_mapped)
        self._dev = self._host_mapped = None

    def resize(self, width, height):
        L = _native.hip()
        self._release()
        self.m_width, self.m_height = max(1, int(width)), max(1, int(height))
        nbytes = self.m_width * self.m_height * 4
        if self.m_type == OutputBufferType.DEVICE:
            p = C.c_void_p()
            _check(self._ctx(), L.pt_device_malloc(self._ctx(), C.byref(p), nbytes), "OutputBuffer.resize")
            self._dev = p.value
        else:
            hp, dp = C.c_void_p(), C.c_void_p()
            _check(self._ctx(), L.pt_host_malloc_mapped(self._ctx(), C.byref(hp), C.byref(dp), nbytes), "OutputBuffer.resize")
            self._host_mapped, self._dev = hp.value, dp.value
        self._host = np.zeros((self.m_height, self.m_width, 4), np.uint8)

    def map(self):
        return self._dev

    def unmap(self):
        pass   # pt_launch returns synchronised (the reference syncs the stream here, CUDAOutputBuffer.h:259-275)

    def getHostPointer(self):
        """uint8 array [height, width, 4]; row 0 is the bottom image row."""
        L = _native.hip()
        nbytes = self.m_width * self.m_height * 4
        if self.m_type == OutputBufferType.DEVICE:
            _check(self._ctx(), L.pt_copy_to_host(self._ctx(), self._host.ctypes.data, self._dev, nbytes), "OutputBuffer.getHostPointer")
        else:
            C.memmove(self._host.ctypes.data, self._host_mapped, nbytes)
        return self._host

    def free(self):
        self._release()


# ------------------------------------------------------------------ render state ----
class PathTracerState:
    """PathTracerState, PathTracerMain.cpp:71-93 (the OptiX handles collapse into one context)."""

    def __init__(self):
        self.context = None
        self.params = PathTraceParams()
        self.refreshAccumulationBuffer = False
        self.frame_counter = 0
        self.sample_summ = 0
        self.total_ms = 0.0
        self._accum_bytes = 0


def createDeviceContext(state, device_id=0, device_ids=None):
    """device_ids (a list): ONE context over several GPUs of the node (pt_create_multi) — pixel tiles per device and an
    RCCL reduce of the accumulation per launch behind the same functions; device_id alone: the reference's one device."""
    L = _native.hip()
    ctx = C.c_void_p()
    if device_ids is not None:
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        rc = L.pt_create_multi(C.byref(ctx), ids, len(device_ids))
    else:
        rc = L.pt_create(C.byref(ctx), int(device_id))
    if rc != 0:
        msg = L.pt_last_error(None)
        raise PathTracerError("createDeviceContext failed: %s" % (msg.decode() if msg else "unknown"))
    state.context = ctx


def buildTheAccelarationStructure(state, objs):
    """Uploads geometry + materials and builds the LBVH on the device."""
    L = _native.hip()
    v = np.ascontiguousarray(objs.getVerticesFloat(), np.float32)
    idx = np.ascontiguousarray(objs.getIndexBuffer(), np.uint32)
    mid = np.ascontiguousarray(objs.getMaterialIndices(), np.uint32)
    mats = objs.getMaterials()
    state._materials = mats
    rc = L.pt_set_scene(state.context, v.ctypes.data, v.size // 4, idx.ctypes.data, idx.size // 3,
                        mid.ctypes.data, C.addressof(mats) if len(mats) else None, len(mats))
    _check(state.context, rc, "buildTheAccelarationStructure")
    state.params.handle = L.pt_scene_handle(state.context)


def createModule(state): pass
def createProgramGroups(state): pass
def createPipeline(state): pass


def createShaderBindingTable(state, obj):
    """The per-material records were uploaded with the scene (pt_set_scene); nothing else to bind."""
    return None


def _alloc_accumulation(state):
    L = _native.hip()
    nbytes = int(state.params.width) * int(state.params.height) * 16
    p = C.c_void_p()
    _check(state.context, L.pt_device_malloc(state.context, C.byref(p), nbytes), "accumulation alloc")
    # zero-filled: under pt_set_partition(rank, world) the pixels of other ranks are never written and the
    # cross-rank reduce(SUM) of distributed.reduce_accumulation relies on them being 0
    _check(state.context, L.pt_device_memset(state.context, p, 0, nbytes), "accumulation clear")
    state.params.accumulationBuffer = p.value
    state._accum_bytes = nbytes


def initializeTheLaunch(state):
    """PathTracerMain.cpp:143-164 — including the hard-coded area light (:154-158)."""
    _alloc_accumulation(state)
    state.params.frameBuffer = None
    state.params.samplesPerPixel = samples_per_launch
    state.params.currentFrameIdx = 0
    al = state.params.areaLight
    al.emission = Float3(10.0, 10.0, 10.0)
    al.corner = Float3(343.0, 547.0, 227.0)
    al.v1 = Float3(0.0, 0.0, 105.0)
    al.v2 = Float3(-130.0, 0.0, 0.0)
    # normalize(cross(v1, v2)) in fp32, vec_math.h:533-549
    v1 = np.array([0.0, 0.0, 105.0], np.float32); v2 = np.array([-130.0, 0.0, 0.0], np.float32)
    c = np.array([v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]], np.float32)
    inv = np.float32(1.0) / np.sqrt(np.float32(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), dtype=np.float32)
    al.normal = _f3(c * inv)


def updateState(output_buffer, state):
    """PathTracerMain.cpp:166-182."""
    if state.refreshAccumulationBuffer:
        state.refreshAccumulationBuffer = False
        state.params.currentFrameIdx = 0
        state.sample_summ = 0
        state.frame_counter = 0
        state.total_ms = 0.0
        L = _native.hip()
        if state.params.accumulationBuffer:
            L.pt_device_free(state.context, state.params.accumulationBuffer)
        _alloc_accumulation(state)


def LaunchCurrentFrame(output_buffer, state, sub_frames=1):
    """PathTracerMain.cpp:184-210: map, launch, unmap, synchronised on return.  sub_frames > 1 renders that many
    consecutive frames (currentFrameIdx, currentFrameIdx + 1, ...) in one kernel launch (pt_launch_frames); the
    buffers end up as after sub_frames separate calls.  The caller advances currentFrameIdx."""
    L = _native.hip()
    state.params.frameBuffer = output_buffer.map() if output_buffer is not None else None
    rc = L.pt_launch_frames(state.context, C.byref(state.params), int(sub_frames))
    if output_buffer is not None:
        output_buffer.unmap()
    _check(state.context, rc, "LaunchCurrentFrame")


def setLightMode(state, mode):
    """0: the reference's estimator (hard-coded rectangle, PathTracerMain.cpp:154-158; the default).  1: the scene's own
    emissive triangles as the area light, light and BSDF sampling combined by the power heuristic (opt-in)."""
    _check(state.context, _native.hip().pt_set_light_mode(state.context, int(mode)), "pt_set_light_mode")
    state.refreshAccumulationBuffer = True


def setMathMode(state, mode):
    """Arithmetic of the shading code (include/acgpt.h pt_set_math_mode).  "fast" / 1 (the default): what the reference's own build
    computes with (nvcc --use_fast_math, CMakeLists.txt:267): approximate reciprocal, square root, sine and cosine.  "ieee" / 0:
    correctly rounded division and square root and the C library's sincosf / acosf, the level the CPU oracle is written at."""
    m = {"ieee": _native.MATH_IEEE, "fast": _native.MATH_FAST}.get(mode, mode)
    _check(state.context, _native.hip().pt_set_math_mode(state.context, int(m)), "pt_set_math_mode")
    state.refreshAccumulationBuffer = True


def getStats(state):
    s = Stats()
    _check(state.context, _native.hip().pt_get_stats(state.context, C.byref(s)), "pt_get_stats")
    return s


def getBvhInfo(state):
    b = BvhInfo()
    _check(state.context, _native.hip().pt_get_bvh_info(state.context, C.byref(b)), "pt_get_bvh_info")
    return b


def readAccumulation(state):
    """float32 [height, width, 4] copy of params.accumulationBuffer (row 0 = bottom)."""
    h, w = int(state.params.height), int(state.params.width)
    out = np.zeros((h, w, 4), np.float32)
    _check(state.context, _native.hip().pt_copy_to_host(state.context, out.ctypes.data, state.params.accumulationBuffer, out.nbytes),
           "readAccumulation")
    return out


def saveAccumulation(state, filename):
    """The progressive state of the reference — params.accumulationBuffer and currentFrameIdx
    (pathTracerPrograms.cu:803-811) — as a file; same format as acgpt_main --save-accum."""
    acc = readAccumulation(state)
    with open(filename, "wb") as fh:
        fh.write(b"ACGPTACC")
        fh.write(np.array([state.params.width, state.params.height, state.params.currentFrameIdx], np.uint32).tobytes())
        fh.write(acc.tobytes())


def restoreAccumulation(state, filename):
    """Continue a saved run: fills params.accumulationBuffer and sets currentFrameIdx to the frames accumulated so far."""
    if state.refreshAccumulationBuffer:      # a pending reset (setMathMode, setLightMode, a key toggle) would discard what is restored here
        updateState(None, state)
    h, w = int(state.params.height), int(state.params.width)
    with open(filename, "rb") as fh:
        blob = fh.read()
    hdr = np.frombuffer(blob[8:20], np.uint32) if len(blob) >= 20 else None
    if blob[:8] != b"ACGPTACC" or hdr is None or int(hdr[0]) != w or int(hdr[1]) != h or len(blob) != 20 + h * w * 16:
        raise PathTracerError("%s is not an accumulation dump of a %dx%d image" % (filename, w, h))
    acc = np.frombuffer(blob[20:], np.float32).copy()
    _check(state.context, _native.hip().pt_copy_to_device(state.context, state.params.accumulationBuffer, acc.ctypes.data, acc.nbytes),
           "restoreAccumulation")
    state.params.currentFrameIdx = int(hdr[2])


def saveImage(filename, rgba):
    """sutil::saveImage conventions (sutil/sutil.cpp:542-655): `rgba` is uint8 [height, width, 4] with row 0
    at the BOTTOM; the file is written top-down; alpha is dropped.  Suffix .ppm or .png."""
    a = np.ascontiguousarray(rgba, np.uint8)
    h, w = a.shape[0], a.shape[1]
    if _native.host().pth_save_image(os.fsencode(filename), a.ctypes.data, w, h) != 0:
        raise PathTracerError("cannot write %s" % filename)


def keyCallback(state, key):
    """PathTracerMain.cpp:100-141.  key: '0' direct lighting, '1' importance sampling,
    'UP' / 'DOWN' max depth +-1 clamped to [1, 28], 'R' reset.  Every change resets accumulation."""
    p = state.params
    if key == "0":
        p.useDirectLighting = 0 if p.useDirectLighting else 1
        state.refreshAccumulationBuffer = True
    elif key == "1":
        p.useImportanceSampling = 0 if p.useImportanceSampling else 1
        state.refreshAccumulationBuffer = True
    elif key == "UP":
        p.maxDepth = min(maxiumumRecursionDepth, int(p.maxDepth) + 1)
        state.refreshAccumulationBuffer = True
    elif key == "DOWN":
        p.maxDepth = max(1, int(p.maxDepth) - 1)
        state.refreshAccumulationBuffer = True
    elif key == "R":
        state.refreshAccumulationBuffer = True


def CleanAllTheThings(state):
    """PathTracerMain.cpp:629-646."""
    L = _native.hip()
    if state.context:
        if state.params.accumulationBuffer:
            L.pt_device_free(state.context, state.params.accumulationBuffer)
            state.params.accumulationBuffer = None
        L.pt_destroy(state.context)
        state.context = None


def setup(obj_path, width=512, height=512, max_depth=4, direct_lighting=False, importance_sampling=False,
          spp=samples_per_launch, device_id=0, build_mode=None, device_ids=None, math_mode=None):
    """The body of main() up to the frame loop (PathTracerMain.cpp:650-684) as one call.  math_mode: None (the library's default,
    "fast"), "ieee" or "fast" (setMathMode)."""
    obj = TinyObjWrapper(obj_path)
    if not obj.dataLoaded:
        raise PathTracerError("cannot load %s: %s" % (obj_path, obj.error))
    state = PathTracerState()
    state.params.width, state.params.height = int(width), int(height)
    state.params.useDirectLighting = 1 if direct_lighting else 0
    state.params.useImportanceSampling = 1 if importance_sampling else 0
    state.params.maxDepth = int(max_depth)
    cam = initCamera()
    cam.setAspectRatio(np.float32(width) / np.float32(height))
    state.params.cameraEye = _f3(cam.eye())
    U, V, W = cam.UVWFrame()
    state.params.cameraU, state.params.cameraV, state.params.cameraW = _f3(U), _f3(V), _f3(W)
    createDeviceContext(state, device_id, device_ids)
    if build_mode is not None:
        _check(state.context, _native.hip().pt_set_build_mode(state.context, int(build_mode)), "pt_set_build_mode")
    buildTheAccelarationStructure(state, obj)
    createModule(state); createProgramGroups(state); createPipeline(state)
    createShaderBindingTable(state, obj)
    initializeTheLaunch(state)
    state.params.samplesPerPixel = int(spp)
    if math_mode is not None:
        setMathMode(state, math_mode)
    return state, obj
