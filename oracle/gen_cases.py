"""Configuration tuples shared by the fixture generator and the tests.  TEST INFRASTRUCTURE ONLY."""

# name, fusion, nb, pos, heads, head_dim, ffn, layers, norm_first, agg, normalize, adapt
# the contrastive configs as shipped (configs/cl_pretrain/*.yaml): raw_encoder_output, 'str_center_uni' views, one bottleneck
# token, the parser's default fusion transformer (4 x 128, ffn 512, 3 layers, post-norm) -- built but never trained there
CL_CASE = ("cl_shipped", "transformer_uni_proj", 1, "learnable", 4, 128, 512, 3, False, "x-attn", False, False)
