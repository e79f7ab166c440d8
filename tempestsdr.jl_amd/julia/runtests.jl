# runtests.jl -- replay tests/golden/v2 through the TempestHIP shim (needs Julia, libtempest_hip.so and an MI355X).
#
#     julia tempestsdr.jl_amd/julia/runtests.jl
#
# Expected values: tests/golden/v2/julia (outputs of the reference itself, written by tests/golden/make_golden.jl)
# when present, else tests/golden/v2/oracle (the CPU restatement).  Bit-exact on the frame path, the tolerances
# of tests/test_julia_golden.py on the FFT paths.
#
# STATUS: written without a Julia runtime (none exists in the build container) -- never executed.
using Test

include(joinpath(@__DIR__, "TempestHIP.jl"))
using .TempestHIP

const V2 = normpath(joinpath(@__DIR__, "..", "..", "tests", "golden", "v2"))
const DT = Dict("f32" => Float32, "f64" => Float64, "c64" => ComplexF32, "i32" => Int32, "i64" => Int64, "u64" => UInt64)

function load_all(dir)
    d = Dict{String,Any}()
    for fn in readdir(dir)
        endswith(fn, ".bin") || continue
        parts = split(fn, ".")
        shape = Tuple(parse.(Int, split(parts[3], "x")))
        a = Array{DT[String(parts[2])]}(undef, shape...)
        open(io -> read!(io, a), joinpath(dir, fn))
        d[String(parts[1])] = a
    end
    return d
end

bits(a) = reinterpret(UInt32, vec(collect(Float32, a)))
samebits(a, b) = size(a) == size(b) && bits(a) == bits(b)
relmax(a, b) = maximum(abs.(Float64.(a) .- Float64.(b))) / maximum(abs.(Float64.(b)))

inp = load_all(joinpath(V2, "inputs"))
refdir = isdir(joinpath(V2, "julia")) && !isempty(readdir(joinpath(V2, "julia"))) ? joinpath(V2, "julia") : joinpath(V2, "oracle")
ref = load_all(refdir)
@info "expected values from $refdir"

@testset "TempestHIP vs $(basename(refdir))" begin
    z = inp["iq"]
    @test samebits(amDemod(z), ref["am"])
    @test samebits(invert_amDemod(z), ref["inv_am"])
    @test maximum(abs.(fmDemod(z) .- ref["fm"])) <= 1f-6
    x = inp["rs_in"]
    @test samebits(sig_to_image(x, 30, 40), ref["s2i"])
    @test samebits(sig_to_image(view(x, 1:333), 30, 40), ref["s2i"])          # contiguous view, no copy (GUI.jl:166)
    big = downgradeImage(inp["img_in"])
    @test size(big) == (600, 800)
    @test samebits(vec(big)[1:997:end], ref["down_sub"])
    nv = zeros(Float32, 375); naiveResampler(nv, inp["up_in"], 3)
    @test samebits(nv, ref["naive"])
    resampler! = init_resampler(Float32, 125, 4)
    up = zeros(Float32, 500); resampler!(up, inp["up_in"])
    @test relmax(up, ref["up_out"]) < 1e-5
    @test_throws AssertionError resampler!(up, zeros(Float32, 124))             # Resampler.jl:47
    # vsync: three calls on one SyncXY (stale s_y), indices identical
    sync = SyncXY(zeros(Float32, 77, 131))
    idx = [vsync(inp["vs_img$(k)"], sync) for k in 0:2]
    @test [i[1] for i in idx] == Int.(ref["vs_idx"][:, 1])
    @test [i[2] for i in idx] == Int.(ref["vs_idx"][:, 2])
    @test idx[1][1] == 1
    # autocorrelation / zoom
    Γ, lags = calculate_autocorrelation(inp["ac_x"], 30000.0, 0.0, 0.05)
    @test maximum(abs.(Γ .- ref["ac_db"])) < 2e-4
    @test length(lags) == length(Γ)
    rates, Γz = zoom_autocorr(Γ, 30000.0; rate_min=25, rate_max=90)
    @test collect(rates) ≈ ref["zoom_rates"]
    @test_throws BoundsError calculate_autocorrelation(ones(Float32, 100), 1000.0, 0, 0.5)   # Autocorrelations.jl:33
    # spectra
    _, y = getSpectrum(1.0, inp["sp_x"]; N=1000)
    @test maximum(abs.(y .- ref["sp_db"])) < 2e-3
    _, w = getWelch(1.0, inp["sp_x"]; sizeFFT=256)
    @test maximum(abs.(w .- ref["welch_db"])) < 1e-3
    _, _, m = getWaterfall(1.0, inp["sp_x"]; sizeFFT=128)
    @test eltype(m) == Float64 && size(m) == (128, 15)
    @test relmax(sqrt.(m), sqrt.(ref["wf"])) < 2e-5
    # fused loop body: the library's default mode is TSDR_FAST -> indices equal up to exact ties, pixels within 6e-7
    for tag in ("A", "B")
        g = inp["fr$(tag)_geom"]; S, y_t, x_t = Int(g[1]), Int(g[2]), Int(g[3])
        state = zeros(Float32, 600, 800)
        frames, sidx = hip_frames!(state, inp["fr$(tag)_iq"], SyncXY(state), S, y_t, x_t, 0.1f0)
        @test size(frames, 3) == Int(g[4])
        @test Int.(permutedims(sidx)) == Int.(ref["fr$(tag)_idx"])
        @test relmax(vec(state)[1:499:end], ref["fr$(tag)_state_sub"]) < 6e-7
    end
end
