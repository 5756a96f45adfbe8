# scLENS_hip.jl -- thin `ccall` shim that puts libsclens_hip.so behind scLENS.sclens(pre_df; device_="hip").
#
# NOT EXECUTED in the build image (no `julia` binary there or on the GPU box): kept deliberately thin, every call is a
# 1:1 binding of include/sclens_hip.h; the same C ABI is exercised by sclens_amd/api.py (ctypes) and the test-suite.
# Reference: Mathbiomed/scLENS v2.0.1, src/scLENS.jl. Include after `using scLENS`:
#     include("scLENS_hip.jl");  res = scLENS.sclens(pre_df; device_="hip")
module ScLENSHip

using SparseArrays, Random, StatsBase, Distributions, DataFrames
import scLENS

const LIB = get(ENV, "SCLENS_HIP_LIB", joinpath(@__DIR__, "..", "sclens_amd", "libsclens_hip.so"))

struct HipError <: Exception
    code::Cint
    msg::String
end
check(ctx, rc) = rc == 0 ? nothing : throw(HipError(rc, unsafe_string(ccall((:sclens_hip_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))))

getint_s(s, name) = begin
    v = Ref{Int64}(0)
    ccall((:sclens_hip_session_get_int, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), s, name, v)
    v[]
end

function with_ctx(f, device::Integer=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:sclens_hip_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device)
    rc == 0 || throw(HipError(rc, "sclens_hip_create"))
    try
        return f(h[])
    finally
        ccall((:sclens_hip_destroy, LIB), Cvoid, (Ptr{Cvoid},), h[])
    end
end

csc0(X::SparseMatrixCSC) = (Int64.(X.colptr) .- 1, Int32.(X.rowval) .- Int32(1), Float32.(X.nzval))

# ---- (A) per-call drop-ins: what `device == "hip"` would do inside the five reference functions -------------------
function _wishart_matrix(ctx, X::Matrix{Float32}; dims=1)                      # scLENS.jl:332-361
    N, M = size(X); n = dims == 2 ? M : N
    Y = Matrix{Float32}(undef, n, n)
    GC.@preserve X Y check(ctx, ccall((:sclens_hip_wishart_matrix_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Cint, Ptr{Float32}), ctx, X, N, M, dims, Y))
    Y
end
function _get_eigen(ctx, Y::Matrix{Float32})                                  # scLENS.jl:375-387
    n = size(Y, 1); L = Vector{Float32}(undef, n); V = Matrix{Float32}(undef, n, n)
    rc = GC.@preserve Y L V ccall((:sclens_hip_get_eigen_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float32}), ctx, Y, n, L, V)
    rc == 6 && return scLENS._get_eigen(Float64.(Y); device="cpu")           # NaN -> Float64 CPU redo (:379-381)
    check(ctx, rc); (L, V)
end
function corr_mat(ctx, X::Matrix{Float32}, Y::Matrix{Float32})               # scLENS.jl:363-373
    n, p = size(X); q = size(Y, 2); out = Matrix{Float32}(undef, p, q)
    GC.@preserve X Y out check(ctx, ccall((:sclens_hip_corr_mat_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Float32}, Int64, Ptr{Float32}), ctx, X, n, p, Y, q, out))
    out
end

function get_denoised_df(ctx, inp_obj)                                        # scLENS.jl:889-931 (body of the DataFrame)
    g = permutedims(Matrix{Float32}(inp_obj[:gene_basis][inp_obj[:sig_id], :]))   # M x s column-major = s rows of M genes
    X0 = Matrix{Float32}(inp_obj[:pca_n1][!, 2:end]); N, s = size(X0); M = size(g, 1)
    rv = inp_obj[:rec_vals]; out = Matrix{Float32}(undef, N, M)
    v(k) = Vector{Float64}(vec(rv[k]))
    tgc, m2m, m2s, ntg, cen = v("TGC"), v("mat2_mean"), v("mat2_std"), v("norm_tgc"), v("cent_")
    GC.@preserve g X0 out tgc m2m m2s ntg cen check(ctx, ccall((:sclens_hip_get_denoised_f32, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Float32}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float32}),
        ctx, X0, N, s, g, M, tgc, m2m, m2s, ntg, cen, out))
    out
end

# logn_scale(pre_scale(x)) (scLENS.jl:650-654) as one call: sparse counts in, dense scaled matrix out
function logn_scale(ctx, X::SparseMatrixCSC; centering="mean")
    N, M = size(X); cp, rv, nz = csc0(X); out = Matrix{Float32}(undef, N, M)
    GC.@preserve cp rv nz out check(ctx, ccall((:sclens_hip_scale_csc_f32, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Cint, Cint, Ptr{Float32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx, N, M, cp, rv, nz, centering == "median" ? 1 : 0, 1, out, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    out
end

# preprocess(tmp_df; ...) (scLENS.jl:160-236): QC masks, gene order and the filtered matrix from the device
function preprocess_hip(tmp_df; min_tp_c=0, min_tp_g=0, max_tp_c=Inf, max_tp_g=Inf, min_genes_per_cell=200,
                        max_genes_per_cell=0, min_cells_per_gene=15, mito_percent=5., ribo_percent=0., device=0)
    gene_name = names(tmp_df)[2:end]
    X = SparseMatrixCSC{Float32,Int64}(scLENS.df2sparr(tmp_df)); N, M = size(X); cp, rv, nz = csc0(X)
    mito = UInt8.(occursin.(r"^(?i)mt-.", gene_name)); ribo = UInt8.(occursin.(r"^(?i)RP[SL].", gene_name))
    keep = zeros(UInt8, N); order = zeros(Int64, M); nc = Ref{Int64}(0); ng = Ref{Int64}(0); nnz_ = Ref{Int64}(0)
    fin(x) = isinf(x) ? floatmax(Float64) : Float64(x)
    with_ctx(device) do ctx
        GC.@preserve cp rv nz mito ribo keep order check(ctx, ccall((:sclens_hip_preprocess_csc, LIB), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Ptr{UInt8}, Ptr{UInt8}, Float64, Float64, Float64, Float64,
             Int64, Int64, Int64, Float64, Float64, Ptr{UInt8}, Ptr{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}),
            ctx, N, M, cp, rv, nz, mito, ribo, fin(min_tp_c), fin(min_tp_g), fin(max_tp_c), fin(max_tp_g), min_genes_per_cell,
            max_genes_per_cell, min_cells_per_gene, mito_percent, ribo_percent, keep, order, nc, ng, nnz_))
        (nc[] == 0 || ng[] == 0) && (println("There is no high quality cells and genes"); return nothing)
        ocp = Vector{Int64}(undef, ng[] + 1); orv = Vector{Int32}(undef, nnz_[]); onz = Vector{Float32}(undef, nnz_[])
        GC.@preserve ocp orv onz check(ctx, ccall((:sclens_hip_preprocess_gather, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}), ctx, ocp, orv, onz))
        o_df = DataFrame(SparseMatrixCSC(nc[], ng[], ocp .+ 1, Int64.(orv) .+ 1, onz), gene_name[order[1:ng[]] .+ 1])
        insertcols!(o_df, 1, :cell => tmp_df.cell[keep .== 1])
        o_df
    end
end

# The reference's count of eigenvalues above lambda_c (:539, :541, :580) with the fp32 eigenvalues near the cut re-evaluated in float64 (L ascending; `refine(lo, hi)` returns the
# Rayleigh quotients of the 0-based ascending indices lo .. hi-1). L receives the refined values; returns (k, nL) with the cut
# taken from the top, so the retained set is contiguous whatever order the refined values of a close pair come out in.
function cut_with_guard_band!(L::Vector{Float64}, lambda_c, refine; guard_band=4.0)
    n = length(L)
    band = guard_band * sqrt(n) * 5.96e-8 * L[end]
    near = findall(abs.(L .- lambda_c) .<= band)
    if guard_band > 0 && !isempty(near) && length(near) <= 64
        L[first(near):last(near)] .= refine(first(near) - 1, last(near))
    end
    below = findlast(x -> !(x > lambda_c), L)                                # strict, as in the reference
    k = below === nothing ? n : n - below
    nL = reverse(L[n-k+1:n])                                                  # value q belongs to eigenvector q of signal_vectors(k)
    issorted(L) || sort!(L)
    (k, nL)
end

# ---- (B) device-resident sclens() --------------------------------------------------------------------------------
# The reference's behaviour under failure (example.jl:9-14, scLENS.jl:504-508, :741-745): no device / out of device memory -> the CPU
# path. Any other code is an error of the call and is re-thrown.
function sclens_hip(inp_df; kwargs...)
    try
        return sclens_hip_device(inp_df; kwargs...)
    catch e
        (e isa HipError && e.code in (2, 3)) || rethrow()
        println("(hip) ", e.msg, " -- falling back to device_=\"cpu\"")
        kw = Dict(kwargs); delete!(kw, :device); delete!(kw, :precision); delete!(kw, :keep_warm)
        return scLENS.sclens(inp_df; device_="cpu", kw...)
    end
end

# `precision`: the arithmetic, as `device_` picks it in the reference (scLENS.jl:649): 1 (device_="hip") = large products from two fp16
# pieces per operand on the fp16 matrix cores with fp32 accumulation, 0 (device_="hip-fp32") = every product on the fp32 matrix cores.
# `keep_warm=true` leaves the call's device blocks in the library's pool for a next call of the same shape; the default hands them back
# to the driver when the call returns (sclens_hip_trim), so that whatever runs next on the GPU (apply_umap!, :863-873) finds them free.
function sclens_hip_device(inp_df; th=60, p_step=0.001, n_perturb=20, centering="mean", device=0, precision=1, keep_warm=false)
    if !(centering in ("mean", "median"))
        # the reference's third branch (:655-657) is scaled_gdata(norm_l(scaled_gdata(x, "mean")), "cent") on a dense Float32 copy:
        # the SAME function of x as the mean branch (z-score, equal-norm rows, centred columns), evaluated in Float32
        println("Warning: The specified centering method is not supported in the current algorithm. scLENS will automatically use mean centering.")
        centering = "mean"
    end
    X_ = scLENS.df2sparr(inp_df)                                              # :662
    N, M = size(X_); nm = min(N, M)
    nz_row, nz_col, nz_val = findnz(X_)
    # R1 zero candidates (:668-673), 0-based for the ABI
    sample_idx = [(i, j) for (i, j) in zip(rand(UInt32(1):UInt32(N), length(nz_val)), rand(UInt32(1):UInt32(M), length(nz_val)))]
    nzz_ = setdiff(sample_idx, [(i, j) for (i, j) in zip(nz_row, nz_col)])
    z1 = UInt32[s[1] - 1 for s in nzz_]; z2 = UInt32[s[2] - 1 for s in nzz_]
    X_r = scLENS.df2sparr(scLENS.random_nz(inp_df, rmix=true))                # R2 (:701)
    cp, rv, nz = csc0(X_); rcp, rrv, rnz = csc0(X_r)
    try
    return with_ctx(device) do ctx
        check(ctx, ccall((:sclens_hip_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), ctx, "precision", precision))
        ses = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve cp rv nz z1 z2 check(ctx, ccall((:sclens_hip_session_create, LIB), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Int64, Ptr{UInt32}, Ptr{UInt32}, Ref{Ptr{Cvoid}}),
            ctx, N, M, cp, rv, nz, length(z1), z1, z2, ses))
        s = ses[]
        try
            median_ = centering == "median"                                  # :653-654; rec_vals stays empty there (:697-698)
            check(ctx, ccall((:sclens_hip_session_set_int, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), s, "centering", median_ ? 1 : 0))
            L = Vector{Float64}(undef, nm); Lr = similar(L)
            rec = median_ ? Dict{String,Vector{Float64}}() :
                  Dict("TGC" => zeros(N), "mat2_mean" => zeros(M), "mat2_std" => zeros(M), "norm_tgc" => zeros(N), "cent_" => zeros(M))
            recp(k) = median_ ? Ptr{Float64}(C_NULL) : pointer(rec[k])
            GC.@preserve rcp rrv rnz L Lr rec check(ctx, ccall((:sclens_hip_session_spectrum, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                s, rcp, rrv, rnz, L, Lr, recp("TGC"), recp("mat2_mean"), recp("mat2_std"), recp("norm_tgc"), recp("cent_")))
            L_mp, _, b_min = scLENS._mp_calculation(L, Lr[1:end-1])           # host statistics stay in Julia (:537-538)
            lambda_c, _ = scLENS._tw(L, L_mp)
            # guard band of the hard cut `L .> lambda_c` (:539, :541): eigenvalues within 4 sqrt(n) eps32 lambda_max of it are
            # replaced by float64 Rayleigh quotients of THEIR eigenvectors (0-based, half-open index range for the ABI). The refined
            # values of a near-degenerate pair may come out in the other order, so the cut is taken from the top and stays
            # contiguous: k = the number of leading indices nm, nm-1, ... whose value exceeds lambda_c -- the vectors
            # `signal_vectors(k)` returns -- and nL[q] is the quotient of vector q (the twin of api.cut_with_guard_band)
            k, nL = cut_with_guard_band!(L, lambda_c, (lo, hi) -> begin
                rho = Vector{Float64}(undef, hi - lo)
                GC.@preserve rho check(ctx, ccall((:sclens_hip_session_refine_eigenvalues, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}), s, lo, hi, rho))
                rho
            end)
            nV = Matrix{Float32}(undef, N, k)
            GC.@preserve nV check(ctx, ccall((:sclens_hip_session_signal_vectors, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float32}), s, k, nV))
            mpC_ = scLENS.mp_check(L_mp)
            p_th = mean([maximum(abs.(rand(Normal(0, sqrt(1 / nm)), nm))) for _ = 1:5000])   # R3 (:709-712)
            r = Ref{Int64}(0)
            check(ctx, ccall((:sclens_hip_session_binary_basis, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int64}), s, C_NULL, r))
            n_2 = round(Int, r[] / 2)                                         # :722
            # the split images of the dense Gram products are idle from here on (12 GB at 100 000 x 30 000): back to the library's pool
            check(ctx, ccall((:sclens_hip_release_scratch, LIB), Cint, (Ptr{Cvoid}, Cstring), ctx, "gram"))
            p_ = 0.999; tank_ = zeros(5, 0); d5 = zeros(5); rit = Ref{Int64}(0)
            while true                                                       # :725-761
                nnzidx = Int(round((1 - p_) * M * N))
                if length(z1) < nnzidx; p_ += p_step; break; end
                sple = UInt32.(sample(UInt32(1):UInt32(length(z1)), nnzidx, replace=false) .- 1)   # R4
                GC.@preserve sple d5 check(ctx, ccall((:sclens_hip_session_search_step, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{UInt32}, Int64, Int64, Ptr{Float64}, Ref{Int64}), s, sple, nnzidx, n_2, d5, rit))
                tank_ = hcat(tank_, d5)
                ppj_ = size(tank_, 2) < 5 ? tank_[2, :] : tank_[2, end-4:end]
                if (sum(ppj_ .< p_th) > 4) | (p_ < 0.9); p_ += 4p_step; break; end
                p_ -= p_step
            end
            min_s = k; min_pc = Int(ceil(min_s * 1.5))
            iszero(min_s) && return Dict(:L => L, :L_mp => L_mp, :λ => lambda_c, :cell_id => string.(inp_df.cell))   # :780-784
            # the full eigensolver's and the search statistic's scratch (30-40 GB) are idle during the ensemble, whose partial
            # eigensolver wants 24 GB of images instead
            check(ctx, ccall((:sclens_hip_release_scratch, LIB), Cint, (Ptr{Cvoid}, Cstring), ctx, "eigensolver"))
            check(ctx, ccall((:sclens_hip_release_scratch, LIB), Cint, (Ptr{Cvoid}, Cstring), ctx, "corr"))
            nLt = zeros(min_pc); nc = Ref{Int64}(0)
            # R5: member t draws from its own generator, so that a member can be solved a second time on the same sample (below)
            pseeds = rand(UInt64, n_perturb)
            member(t) = begin                                                # :771-778
                sple = UInt32.(sample(MersenneTwister(pseeds[t]), UInt32(1):UInt32(length(z1)), Int(round((1 - p_) * M * N)), replace=false) .- 1)
                GC.@preserve sple nLt check(ctx, ccall((:sclens_hip_session_perturb, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt32}, Int64, Int64, Ptr{Float64}, Ref{Int64}), s, t - 1, sple, length(sple), min_pc, nLt, nc))
            end
            # the eigenpairs k .. min_pc-1 of a member (:776) are consumed only if the matching (:788) picks one. From order 16 000 they are
            # not converged at all ("chefsi_tail_free"); the library then reports per member whether the matching provably does not
            # depend on them ("match_uncertain:<t>", include/sclens_hip.h), and the members without that proof -- or whose matching
            # picked a tail column -- are solved again with the tail converged
            tail_free = nm >= 16000 && getint_s(s, "chefsi") != 0      # (only where the partial eigensolver is on at all)
            setint(name, v) = check(ctx, ccall((:sclens_hip_session_set_int, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), s, name, v))
            getint(name) = begin
                v = Ref{Int64}(0)
                check(ctx, ccall((:sclens_hip_session_get_int, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), s, name, v))
                v[]
            end
            setint("chefsi_tail_free", tail_free ? 1 : 0)
            for t in 1:n_perturb
                member(t)
            end
            npairs = div(n_perturb * (n_perturb - 1), 2)
            a_b = Matrix{Int32}(undef, k, n_perturb); bt = Matrix{Float64}(undef, npairs, k)   # row-major k x npairs
            score() = GC.@preserve a_b bt check(ctx, ccall((:sclens_hip_session_robustness, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Float64}), s, n_perturb, a_b, bt))
            score()
            # only members the PARTIAL eigensolver produced can gain from a second solve (the full solver would return the same vectors)
            again = getint("chefsi_used") > 0 ?
                    [t for t in 1:n_perturb if any(a_b[:, t] .>= k) || (tail_free && getint("match_uncertain:$(t - 1)") != 0)] : Int[]
            if !isempty(again)
                setint("chefsi_tail_free", 0); setint("chefsi_tail_gap_milli", 50)
                for t in again
                    member(t)
                end
                setint("chefsi_tail_gap_milli", 0)
                score()
            end
            b_ = permutedims(bt)
            q1 = mapslices(x -> quantile(x, 0.25), b_, dims=2)[:]; q3 = mapslices(x -> quantile(x, 0.75), b_, dims=2)[:]
            iq = mapslices(iqr, b_, dims=2)[:]
            filt = [b_[i, :][q1[i]-1.5*iq[i].<=b_[i, :].<=q3[i]+1.5*iq[i]] for i = 1:k]
            m_score = median.(filt); sd_score = std.(filt)
            sig_id = findall(m_score .> cos(deg2rad(th)))
            gt = Matrix{Float32}(undef, M, k)                                 # k rows of M genes, row-major
            GC.@preserve nL gt check(ctx, ccall((:sclens_hip_session_gene_basis, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float32}), s, Float64.(nL), gt))
            df_X0 = DataFrame(nV .* (sqrt.(nL))', :auto); insertcols!(df_X0, 1, :cell => inp_df.cell)
            df_X1 = DataFrame(nV[:, sig_id] .* sqrt.(nL[sig_id])', :auto); insertcols!(df_X1, 1, :cell => inp_df.cell)
            return Dict(:pca => df_X0, :pca_n1 => df_X1, :sig_id => sig_id, :L => L, :L_mp => L_mp, :λ => lambda_c,
                :robustness_scores => Dict(:b_ => b_, :rob_score => m_score, :m_scores => m_score, :sd_scores => sd_score),
                :signal_evec => nV, :signal_ev => nL, :cell_id => inp_df.cell, :gene_id => names(inp_df)[2:end],
                :gene_basis => permutedims(gt), :pass => mpC_[:pass], :rec_vals => rec)   # keys of :826-829
        finally
            ccall((:sclens_hip_session_destroy, LIB), Cvoid, (Ptr{Cvoid},), s)
        end
    end
    finally
        # AFTER with_ctx has destroyed the context (also when the call throws): only then are its named workspaces (24 GB of
        # partial-eigensolver images, the vector blocks, the re-grown eigensolver scratch) back in the pool's idle cache, and the trim
        # hands everything to the driver
        keep_warm || ccall((:sclens_hip_trim, LIB), Cint, (Cint,), device)
    end
end

end # module

# One-line hook a maintainer adds at the top of scLENS.sclens (src/scLENS.jl:649):
#     device_ in ("hip", "hip-fp32") && return ScLENSHip.sclens_hip(inp_df; th=th, p_step=p_step, n_perturb=n_perturb, centering=centering,
#                                                                     precision=(device_ == "hip" ? 1 : 0))
# sclens_hip falls back to device_="cpu" on HipError code 2 (no device) or 3 (out of device memory), mirroring example.jl:9-14 and :504-508.
