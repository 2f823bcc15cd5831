import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import torch, bench
import imageexperiments_amd as ia
for name,(W,H,K,q) in bench.WORKLOADS.items():
    rgb=bench.synth_frame(W,H,12345)
    ctx=ia.create_compression_context(K,8,q,device=0)
    ctx.encode_image(rgb[:64,:64].copy())
    t=time.perf_counter(); counts,choices,en,sw=ctx.encode_tiles(rgb); t1=time.perf_counter()
    blob=ia.assemble_streams(W,H,K,8,ctx.quant,counts,choices.view(np.uint32)); t2=time.perf_counter()
    img=ia.decode_image(blob,ctx); t3=time.perf_counter()
    print(f"{name}: encode_tiles(host buffers, incl. alloc+H2D+D2H) {1e3*(t1-t):.1f} ms, host entropy stage {1e3*(t2-t1):.1f} ms, bytes {len(blob)}, bpp {8*len(blob)/(W*H):.3f}, decode_image(device tiles + host parse) {1e3*(t3-t2):.1f} ms, psnr {ia.calculate_psnr(rgb,img):.2f}")
    ctx.close()
