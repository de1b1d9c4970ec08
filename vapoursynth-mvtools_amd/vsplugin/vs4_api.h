/*
 * vs4_api.h -- the part of the VapourSynth API v4 that a filter plugin binds, declared here because this build image
 * ships neither VapourSynth nor its headers.
 *
 * STATUS: reconstructed from the public API-4 documentation, NOT copied from and NOT verified against the real
 * <VapourSynth4.h>.  Struct member ORDER is ABI.  Before loading libmvtools_vs.so into a real VapourSynth core,
 * build it with -DMVX_USE_SYSTEM_VS_HEADER (then <VapourSynth4.h> is used instead of this file) -- the shim source
 * only uses names that exist in the real header.  The in-repo mini host (minihost.c) shares this file, so the tests
 * are self-consistent either way.
 *
 * Only what the mvtools hot path uses is spelled out (27 VSAPI members + 2 VSPLUGINAPI members); the
 * other slots are kept as untyped placeholders so that the offsets of the used members match the documented table.
 */
#ifndef MVX_VS4_API_H
#define MVX_VS4_API_H

#ifdef MVX_USE_SYSTEM_VS_HEADER
#include <VapourSynth4.h>
#else

#include <stddef.h>
#include <stdint.h>

#define VS_CC
#ifdef __cplusplus
#define VS_EXTERNAL_API(ret) extern "C" __attribute__((visibility("default"))) ret VS_CC
#else
#define VS_EXTERNAL_API(ret) __attribute__((visibility("default"))) ret VS_CC
#endif

#define VS_MAKE_VERSION(major, minor) (((major) << 16) | (minor))
#define VAPOURSYNTH_API_MAJOR 4
#define VAPOURSYNTH_API_MINOR 0
#define VAPOURSYNTH_API_VERSION VS_MAKE_VERSION(VAPOURSYNTH_API_MAJOR, VAPOURSYNTH_API_MINOR)

typedef struct VSFrame VSFrame;
typedef struct VSNode VSNode;
typedef struct VSCore VSCore;
typedef struct VSPlugin VSPlugin;
typedef struct VSPluginFunction VSPluginFunction;
typedef struct VSFunction VSFunction;
typedef struct VSMap VSMap;
typedef struct VSLogHandle VSLogHandle;
typedef struct VSFrameContext VSFrameContext;
typedef struct VSAPI VSAPI;
typedef struct VSPLUGINAPI VSPLUGINAPI;

typedef enum VSColorFamily { cfUndefined = 0, cfGray = 1, cfRGB = 2, cfYUV = 3 } VSColorFamily;
typedef enum VSSampleType { stInteger = 0, stFloat = 1 } VSSampleType;
typedef enum VSFilterMode { fmParallel = 0, fmParallelRequests = 1, fmUnordered = 2, fmFrameState = 3 } VSFilterMode;
typedef enum VSMediaType { mtVideo = 1, mtAudio = 2 } VSMediaType;
typedef enum VSPropertyType { ptUnset = 0, ptInt = 1, ptFloat = 2, ptData = 3, ptFunction = 4, ptVideoNode = 5, ptAudioNode = 6, ptVideoFrame = 7, ptAudioFrame = 8 } VSPropertyType;
typedef enum VSMapPropertyError { peSuccess = 0, peUnset = 1, peType = 2, peIndex = 4, peError = 3 } VSMapPropertyError;
typedef enum VSMapAppendMode { maReplace = 0, maAppend = 1 } VSMapAppendMode;
typedef enum VSActivationReason { arError = -1, arInitial = 0, arAllFramesReady = 1 } VSActivationReason;
typedef enum VSDataTypeHint { dtUnknown = -1, dtBinary = 0, dtUtf8 = 1 } VSDataTypeHint;
typedef enum VSRequestPattern { rpGeneral = 0, rpNoFrameReuse = 1, rpStrictSpatial = 2 } VSRequestPattern;

typedef struct VSVideoFormat {
    int colorFamily;
    int sampleType;
    int bitsPerSample;
    int bytesPerSample;
    int subSamplingW;
    int subSamplingH;
    int numPlanes;
} VSVideoFormat;

typedef struct VSVideoInfo {
    VSVideoFormat format;
    int64_t fpsNum;
    int64_t fpsDen;
    int width;
    int height;
    int numFrames;
} VSVideoInfo;

typedef struct VSFilterDependency {
    VSNode *source;
    int requestPattern;
} VSFilterDependency;

typedef void (VS_CC *VSPublicFunction)(const VSMap *in, VSMap *out, void *userData, VSCore *core, const VSAPI *vsapi);
typedef void (VS_CC *VSInitPlugin)(VSPlugin *plugin, const VSPLUGINAPI *vspapi);
typedef const VSFrame *(VS_CC *VSFilterGetFrame)(int n, int activationReason, void *instanceData, void **frameData, VSFrameContext *frameCtx, VSCore *core, const VSAPI *vsapi);
typedef void (VS_CC *VSFilterFree)(void *instanceData, VSCore *core, const VSAPI *vsapi);

struct VSPLUGINAPI {
    int (VS_CC *getAPIVersion)(void);
    int (VS_CC *configPlugin)(const char *identifier, const char *pluginNamespace, const char *name, int pluginVersion, int apiVersion, int flags, VSPlugin *plugin);
    int (VS_CC *registerFunction)(const char *name, const char *args, const char *returnType, VSPublicFunction argsFunc, void *functionData, VSPlugin *plugin);
};

typedef void (*mvx_vs_slot)(void); /* placeholder for members this path never calls */

struct VSAPI {
    /* filters and nodes */
    void (VS_CC *createVideoFilter)(VSMap *out, const char *name, const VSVideoInfo *vi, VSFilterGetFrame getFrame, VSFilterFree free, int filterMode, const VSFilterDependency *dependencies, int numDeps, void *instanceData, VSCore *core);
    mvx_vs_slot createVideoFilter2, createAudioFilter, createAudioFilter2, setLinearFilter, setCacheMode, setCacheOptions;
    void (VS_CC *freeNode)(VSNode *node);
    VSNode *(VS_CC *addNodeRef)(VSNode *node);
    mvx_vs_slot getNodeType;
    const VSVideoInfo *(VS_CC *getVideoInfo)(VSNode *node);
    mvx_vs_slot getAudioInfo;

    /* frames */
    VSFrame *(VS_CC *newVideoFrame)(const VSVideoFormat *format, int width, int height, const VSFrame *propSrc, VSCore *core);
    mvx_vs_slot newVideoFrame2, newAudioFrame, newAudioFrame2;
    void (VS_CC *freeFrame)(const VSFrame *f);
    mvx_vs_slot addFrameRef;
    VSFrame *(VS_CC *copyFrame)(const VSFrame *f, VSCore *core);
    const VSMap *(VS_CC *getFramePropertiesRO)(const VSFrame *f);
    VSMap *(VS_CC *getFramePropertiesRW)(VSFrame *f);
    ptrdiff_t (VS_CC *getStride)(const VSFrame *f, int plane);
    const uint8_t *(VS_CC *getReadPtr)(const VSFrame *f, int plane);
    uint8_t *(VS_CC *getWritePtr)(VSFrame *f, int plane);
    const VSVideoFormat *(VS_CC *getVideoFrameFormat)(const VSFrame *f);
    mvx_vs_slot getAudioFrameFormat, getFrameType;
    int (VS_CC *getFrameWidth)(const VSFrame *f, int plane);
    int (VS_CC *getFrameHeight)(const VSFrame *f, int plane);
    mvx_vs_slot getFrameLength;

    /* formats */
    mvx_vs_slot getVideoFormatName, getAudioFormatName, queryVideoFormat, queryAudioFormat, queryVideoFormatID, getVideoFormatByID;

    /* frame requests */
    const VSFrame *(VS_CC *getFrame)(int n, VSNode *node, char *errorMsg, int bufSize);
    mvx_vs_slot getFrameAsync;
    const VSFrame *(VS_CC *getFrameFilter)(int n, VSNode *node, VSFrameContext *frameCtx);
    void (VS_CC *requestFrameFilter)(int n, VSNode *node, VSFrameContext *frameCtx);
    mvx_vs_slot releaseFrameEarly, cacheFrame;
    void (VS_CC *setFilterError)(const char *errorMessage, VSFrameContext *frameCtx);

    /* external functions */
    mvx_vs_slot createFunction, freeFunction, addFunctionRef, callFunction;

    /* maps */
    VSMap *(VS_CC *createMap)(void);
    void (VS_CC *freeMap)(VSMap *map);
    void (VS_CC *clearMap)(VSMap *map);
    mvx_vs_slot copyMap;
    void (VS_CC *mapSetError)(VSMap *map, const char *errorMessage);
    const char *(VS_CC *mapGetError)(const VSMap *map);
    mvx_vs_slot mapNumKeys, mapGetKey, mapDeleteKey;
    int (VS_CC *mapNumElements)(const VSMap *map, const char *key);
    mvx_vs_slot mapGetType, mapSetEmpty;
    int64_t (VS_CC *mapGetInt)(const VSMap *map, const char *key, int index, int *error);
    int (VS_CC *mapGetIntSaturated)(const VSMap *map, const char *key, int index, int *error);
    mvx_vs_slot mapGetIntArray;
    int (VS_CC *mapSetInt)(VSMap *map, const char *key, int64_t i, int append);
    mvx_vs_slot mapSetIntArray;
    double (VS_CC *mapGetFloat)(const VSMap *map, const char *key, int index, int *error);
    mvx_vs_slot mapGetFloatSaturated, mapGetFloatArray;
    int (VS_CC *mapSetFloat)(VSMap *map, const char *key, double d, int append);
    mvx_vs_slot mapSetFloatArray;
    const char *(VS_CC *mapGetData)(const VSMap *map, const char *key, int index, int *error);
    int (VS_CC *mapGetDataSize)(const VSMap *map, const char *key, int index, int *error);
    mvx_vs_slot mapGetDataTypeHint;
    int (VS_CC *mapSetData)(VSMap *map, const char *key, const char *data, int size, int type, int append);
    VSNode *(VS_CC *mapGetNode)(const VSMap *map, const char *key, int index, int *error);
    int (VS_CC *mapSetNode)(VSMap *map, const char *key, VSNode *node, int append);
    mvx_vs_slot mapConsumeNode, mapGetFrame, mapSetFrame, mapConsumeFrame, mapGetFunction, mapSetFunction, mapConsumeFunction;

    /* plugins */
    mvx_vs_slot registerFunction;
    VSPlugin *(VS_CC *getPluginByID)(const char *identifier, VSCore *core);
    mvx_vs_slot getPluginByNamespace, getNextPlugin, getPluginName, getPluginID, getPluginNamespace,
        getNextPluginFunction, getPluginFunctionByName, getPluginFunctionName, getPluginFunctionArguments, getPluginFunctionReturnType,
        getPluginPath, getPluginVersion;
    VSMap *(VS_CC *invoke)(VSPlugin *plugin, const char *name, const VSMap *args);

    /* core */
    mvx_vs_slot createCore, freeCore, setMaxCacheSize, setThreadCount, getCoreInfo, getAPIVersion;

    /* logging */
    void (VS_CC *logMessage)(int msgType, const char *msg, VSCore *core);
    mvx_vs_slot addLogHandler, removeLogHandler;
};

#endif /* MVX_USE_SYSTEM_VS_HEADER */

/* VSHelper4.h equivalents used by the shim */
static inline int mvx_vsh_is_constant_video_format(const VSVideoInfo *vi) {
    return vi->height > 0 && vi->width > 0 && vi->format.colorFamily != cfUndefined;
}

#endif /* MVX_VS4_API_H */
