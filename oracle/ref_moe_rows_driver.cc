// ref_moe_rows_driver.cc -- TEST INFRASTRUCTURE (oracle/_ref): the reference's own per-row expert selection,
// HostTensorOpr::BuildRowsForMoE (src/tensor/host_tensor_opr.cc:190-244, compiled where it lies by oracle/Makefile's
// `ref_moe_rows` target), on a [tokens][experts] F16 probability matrix.  Used by tests/golden/gen_moe_rows_fixtures.py to pin
// oracle.moe_topk and the device routing kernel (ifa_moe_route_topk) to the reference.  Nothing in inferflow_amd/ links this.
//
//   ifa_ref_moe_rows <in.bin> <out.bin>
// in.bin : int32 magic 0x49464d31, tokens, experts, top_k, norm, then uint16 probs[tokens][experts] (F16 bits)
// out.bin: per token int32 size, then 8 x (int32 expert, float32 weight) in the reference's order (unused slots: -1, 0)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "tensor/host_tensor_opr.h"

using namespace inferflow;

int main(int argc, const char *argv[])
{
    if (argc < 3) { fprintf(stderr, "usage: %s <in.bin> <out.bin>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    int32_t head[5];
    if (!f || fread(head, 4, 5, f) != 5 || head[0] != 0x49464d31) { fprintf(stderr, "bad input\n"); return 2; }
    const int T = head[1], E = head[2], K = head[3], norm = head[4];
    std::vector<uint16_t> bits((size_t)T * E);
    if (fread(bits.data(), 2, bits.size(), f) != bits.size()) { fprintf(stderr, "short input\n"); return 2; }
    fclose(f);
    HostTensor probs;
    probs.New(ElementType::F16, E, T);
    memcpy((void *)probs.data_f16(), bits.data(), bits.size() * 2);
    std::vector<RowItemForMoe> rows;
    HostTensorOpr::BuildRowsForMoE(rows, probs, K, norm != 0);
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    for (int t = 0; t < T; t++) {
        int32_t n = rows[t].size;
        fwrite(&n, 4, 1, o);
        for (int i = 0; i < RowItemForMoe::MAX_SIZE; i++) {
            int32_t e = i < n ? (int32_t)rows[t].arr[i].id : -1;
            float w = i < n ? rows[t].arr[i].weight : 0.0f;
            fwrite(&e, 4, 1, o); fwrite(&w, 4, 1, o);
        }
    }
    fclose(o);
    return 0;
}
