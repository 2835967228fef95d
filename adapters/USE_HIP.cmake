# USE_HIP.cmake -- the switch a maintainer of ZhenghaoFei/visual_odom adds next to the existing USE_CUDA one
# (CMakeLists.txt:5-9: `option(USE_CUDA ...)` / `add_definitions(-DUSE_CUDA)`).
#
#   top-level CMakeLists.txt, after the USE_CUDA block:
#       set(VO_HIP_ROOT "" CACHE PATH "checkout of the MI355X front end (holds include/vo_hip.h, adapters/, visual_odom_amd/libvo_hip.so)")
#       include(${VO_HIP_ROOT}/adapters/USE_HIP.cmake)
#   src/CMakeLists.txt, after the executable target `vo` is defined:
#       vo_use_hip(vo)
#   cmake -DUSE_HIP=ON -DVO_HIP_ROOT=/path/to/this/repo ..
#
# and the three call sites (INTEGRATION.md shows them): visualOdometry.cpp:112-118 gets an `#if USE_HIP` branch calling
# circularMatching_hip(...) in front of the `#if USE_CUDA` one; main.cpp:169-171 calls triangulate_hip(...); main.cpp:181
# calls trackingFrame2Frame_hip(...).  libvo_hip.so is built by `python -m visual_odom_amd.build` (hipcc, gfx950).
option(USE_HIP "Run circularMatching / triangulation / PnP on an AMD MI355X through libvo_hip.so" OFF)

if(USE_HIP)
    if(NOT VO_HIP_ROOT)
        message(FATAL_ERROR "USE_HIP=ON needs -DVO_HIP_ROOT=<checkout of the MI355X front end>")
    endif()
    add_definitions(-DUSE_HIP)
    find_library(VO_HIP_LIBRARY vo_hip PATHS ${VO_HIP_ROOT}/visual_odom_amd NO_DEFAULT_PATH)
    if(NOT VO_HIP_LIBRARY)
        message(FATAL_ERROR "libvo_hip.so not found under ${VO_HIP_ROOT}/visual_odom_amd: run `python -m visual_odom_amd.build` there")
    endif()
endif()

function(vo_use_hip target)
    if(USE_HIP)
        target_sources(${target} PRIVATE ${VO_HIP_ROOT}/adapters/feature_hip.cpp)
        target_include_directories(${target} PRIVATE ${VO_HIP_ROOT}/include ${VO_HIP_ROOT}/adapters)
        target_link_libraries(${target} ${VO_HIP_LIBRARY})
        set_target_properties(${target} PROPERTIES BUILD_RPATH "${VO_HIP_ROOT}/visual_odom_amd;/opt/rocm/lib")
    endif()
endfunction()
