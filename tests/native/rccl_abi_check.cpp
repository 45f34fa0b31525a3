// rccl_abi_check.cpp -- compile-time proof that the function-pointer types librfhip.so calls RCCL
// through (reforge_amd/csrc/rf_rccl_abi.h; the same types tests/native/fake_rccl.cpp implements)
// are ABI-equivalent to the prototypes of the REAL library's header, and a run-time check that the
// real librccl exports each symbol.  Built and run by tests/test_rccl_abi.py on the CPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <type_traits>

#include "rf_rccl_abi.h"

namespace {

// two parameter/return types pass identically in the x86-64 SysV ABI if both are pointers, or both
// integral/enum of one size, or both trivially copyable classes of one size and alignment
template <class A, class B> struct abi_same {
    static constexpr bool ptr = std::is_pointer<A>::value && std::is_pointer<B>::value;
    static constexpr bool num = (std::is_integral<A>::value || std::is_enum<A>::value) &&
                                (std::is_integral<B>::value || std::is_enum<B>::value) && sizeof(A) == sizeof(B);
    static constexpr bool cls = std::is_class<A>::value && std::is_class<B>::value && sizeof(A) == sizeof(B) &&
                                alignof(A) == alignof(B) && std::is_trivially_copyable<A>::value && std::is_trivially_copyable<B>::value;
    static constexpr bool value = ptr || num || cls;
};

template <class F, class G> struct fn_same : std::false_type {};
template <class R1, class... A1, class R2, class... A2> struct fn_same<R1 (*)(A1...), R2 (*)(A2...)> {
    template <bool SameCount, class Dummy = void> struct args : std::false_type {};
    template <class Dummy> struct args<true, Dummy> {
        static constexpr bool value = (abi_same<A1, A2>::value && ... && true);
    };
    static constexpr bool value = abi_same<R1, R2>::value && args<sizeof...(A1) == sizeof...(A2)>::value;
};

static_assert(sizeof(ncclUniqueId) == sizeof(rf::NcclId) && alignof(ncclUniqueId) == alignof(rf::NcclId), "ncclUniqueId is 128 bytes");
static_assert((int)ncclInt8 == rf::kNcclChar && (int)ncclChar == rf::kNcclChar, "ncclInt8 == 0");
static_assert((int)ncclSuccess == rf::kNcclSuccess, "ncclSuccess == 0");
static_assert(fn_same<decltype(&ncclGetUniqueId), rf::NcclGetUniqueIdFn>::value, "ncclGetUniqueId");
static_assert(fn_same<decltype(&ncclCommInitRank), rf::NcclCommInitRankFn>::value, "ncclCommInitRank");
static_assert(fn_same<decltype(&ncclCommDestroy), rf::NcclCommDestroyFn>::value, "ncclCommDestroy");
static_assert(fn_same<decltype(&ncclSend), rf::NcclSendFn>::value, "ncclSend");
static_assert(fn_same<decltype(&ncclRecv), rf::NcclRecvFn>::value, "ncclRecv");
static_assert(fn_same<decltype(&ncclGroupStart), rf::NcclGroupFn>::value, "ncclGroupStart");
static_assert(fn_same<decltype(&ncclGroupEnd), rf::NcclGroupFn>::value, "ncclGroupEnd");
static_assert(fn_same<decltype(&ncclGetErrorString), rf::NcclGetErrorStringFn>::value, "ncclGetErrorString");
// a deliberately wrong type must be rejected, or the checks above prove nothing
static_assert(!fn_same<decltype(&ncclSend), rf::NcclGroupFn>::value, "the checker can fail");
static_assert(!fn_same<decltype(&ncclSend), int (*)(const void*, int, int, int, void*, hipStream_t)>::value, "size_t vs int is caught");

}  // namespace

int main(int argc, char** argv)
{
    // which library would the product's dlopen("librccl.so.1") map, and does it export the eight symbols?
    const char* name = argc > 1 ? argv[1] : "librccl.so.1";
    void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::fprintf(stderr, "dlopen(%s): %s\n", name, dlerror()); return 2; }
    const char* syms[] = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"};
    for (const char* s : syms) {
        void* p = dlsym(h, s);
        if (!p) { std::fprintf(stderr, "%s lacks %s\n", name, s); return 3; }
        Dl_info info;
        if (dladdr(p, &info) && info.dli_fname) std::printf("%s %s\n", s, info.dli_fname);
    }
    return 0;
}
