/*
 * vs4_api_check.h -- compile-time statement of the VSAPI layout the PREBUILT shell binds.
 *
 * libmvtools_vs.so in this repository is compiled against the reconstructed vs4_api.h (the build image has no VapourSynth headers).
 * Each line below pins the slot (pointer-sized index inside `struct VSAPI`) of one member the shell calls, as that header lays it
 * out.  Built with -DMVX_USE_SYSTEM_VS_HEADER the same asserts run against the real <VapourSynth4.h>: if the reconstruction
 * disagrees with the real header anywhere the shell depends on, THAT build fails here, loudly, naming the member -- which is
 * the check that the prebuilt library could not have been given inside this image.
 */
#ifndef MVX_VS4_API_CHECK_H
#define MVX_VS4_API_CHECK_H
#include <stddef.h>
#define MVX_VSAPI_SLOT(member, slot) _Static_assert(offsetof(VSAPI, member) == (size_t)(slot) * sizeof(void *), "VSAPI layout differs from vs4_api.h at member " #member)
MVX_VSAPI_SLOT(addNodeRef, 8);
MVX_VSAPI_SLOT(copyFrame, 18);
MVX_VSAPI_SLOT(createMap, 47);
MVX_VSAPI_SLOT(createVideoFilter, 0);
MVX_VSAPI_SLOT(freeFrame, 16);
MVX_VSAPI_SLOT(freeMap, 48);
MVX_VSAPI_SLOT(freeNode, 7);
MVX_VSAPI_SLOT(getFrame, 36);
MVX_VSAPI_SLOT(getFrameFilter, 38);
MVX_VSAPI_SLOT(getFrameHeight, 28);
MVX_VSAPI_SLOT(getFramePropertiesRO, 19);
MVX_VSAPI_SLOT(getFramePropertiesRW, 20);
MVX_VSAPI_SLOT(getFrameWidth, 27);
MVX_VSAPI_SLOT(getPluginByID, 83);
MVX_VSAPI_SLOT(getReadPtr, 22);
MVX_VSAPI_SLOT(getStride, 21);
MVX_VSAPI_SLOT(getVideoInfo, 10);
MVX_VSAPI_SLOT(getWritePtr, 23);
MVX_VSAPI_SLOT(invoke, 96);
MVX_VSAPI_SLOT(mapGetData, 69);
MVX_VSAPI_SLOT(mapGetDataSize, 70);
MVX_VSAPI_SLOT(mapGetError, 52);
MVX_VSAPI_SLOT(mapGetFloat, 64);
MVX_VSAPI_SLOT(mapGetInt, 59);
MVX_VSAPI_SLOT(mapGetIntSaturated, 60);
MVX_VSAPI_SLOT(mapGetNode, 73);
MVX_VSAPI_SLOT(mapSetData, 72);
MVX_VSAPI_SLOT(mapSetError, 51);
MVX_VSAPI_SLOT(mapSetInt, 62);
MVX_VSAPI_SLOT(mapSetNode, 74);
MVX_VSAPI_SLOT(newVideoFrame, 12);
MVX_VSAPI_SLOT(requestFrameFilter, 39);
MVX_VSAPI_SLOT(setFilterError, 42);
_Static_assert(offsetof(VSPLUGINAPI, configPlugin) == sizeof(void *) && offsetof(VSPLUGINAPI, registerFunction) == 2 * sizeof(void *), "VSPLUGINAPI layout differs from vs4_api.h");
_Static_assert(sizeof(VSVideoFormat) == 7 * sizeof(int) && offsetof(VSVideoInfo, fpsNum) == 32 && offsetof(VSVideoInfo, width) == 48 && offsetof(VSVideoInfo, numFrames) == 56, "VSVideoInfo layout differs from vs4_api.h");
_Static_assert(sizeof(VSFilterDependency) == 2 * sizeof(void *) && offsetof(VSFilterDependency, requestPattern) == sizeof(void *), "VSFilterDependency layout differs from vs4_api.h");
_Static_assert(arInitial == 0 && arAllFramesReady == 1 && arError == -1 && fmParallel == 0 && maReplace == 0 && dtBinary == 0 && cfGray == 1 && cfYUV == 3 && stInteger == 0 && rpGeneral == 0 && rpStrictSpatial == 2,
               "enum values differ from vs4_api.h");
#endif
