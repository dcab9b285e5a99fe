"""CPU: the residency rule behind every launch that waits for sibling workgroups INSIDE the launch (fused QKV + attention, split-K
halves of the large-tile GEMM, K parts of the rows GEMM) -- VERDICT r4 item 8, ADVICE r4: such a launch is chosen only when its
whole grid can be resident at once on the CUs this process may use; the CU count is mocked here (no device needed)."""
import inferflow_amd as ia


def test_grid_must_fit_occupancy_times_visible_cus():
    L = ia.lib()
    # 256 CUs, one workgroup per CU: the 256-workgroup fused launch fits, 258 K-parts workgroups do not
    assert L.ifa_wait_grid_decision(1, 256, 256) == 1
    assert L.ifa_wait_grid_decision(1, 256, 258) == 0
    # two workgroups per CU (split-K halves of 128 x 128 tiles): 512 fit, 513 do not
    assert L.ifa_wait_grid_decision(2, 256, 512) == 1 and L.ifa_wait_grid_decision(2, 256, 513) == 0
    # a CU mask that leaves 128 CUs: the same 256-workgroup grid must not be chosen (half of it would queue behind the other half
    # and the resident half would wait for it until the timeout)
    assert L.ifa_wait_grid_decision(1, 128, 256) == 0 and L.ifa_wait_grid_decision(1, 128, 128) == 1
    # a kernel the occupancy calculator cannot place at all never waits
    assert L.ifa_wait_grid_decision(0, 256, 1) == 0


def test_cu_masks_are_read():
    L = ia.lib()
    f = L.ifa_visible_cus_from_mask
    assert f(None, 0, 256) == 256 and f(b"", 0, 256) == 256
    assert f(b"0xffffffff", 0, 256) == 32                       # ROC_GLOBAL_CU_MASK: a hex mask, one bit per CU
    assert f(b"ffff", 0, 256) == 16
    assert f(b"0:0-31", 0, 256) == 32                           # HSA_CU_MASK: <gpu>:<ranges>
    assert f(b"0:0-31,64-95;1:0-7", 0, 256) == 64 and f(b"0:0-31,64-95;1:0-7", 1, 256) == 8
    assert f(b"1:0-7", 0, 256) == 256                           # no entry for this device: everything
    assert f(b"garbage", 0, 256) == 256                         # unreadable: assume the device's own count
    assert f(b"0:0-511", 0, 256) == 256                         # never more than the device has
    # a GPU LIST in front of the colon (ids and ranges): every listed device gets the mask
    assert f(b"0,1:0-31", 1, 256) == 32 and f(b"0,2-3:0-15", 3, 256) == 16 and f(b"0,2-3:0-15", 1, 256) == 256
    # a mask that names GPUs in a form that cannot be read: conservative -- 0 CUs, so no launch that waits is chosen
    assert f(b"gpu0:0-31", 0, 256) == 0


def test_waits_start_enabled():
    assert ia.lib().ifa_inlaunch_waits_enabled() in (0, 1)
